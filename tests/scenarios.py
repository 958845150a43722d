"""Scripted decode / KV-op sequences shared by the golden generator (tests/golden/make_golden.py), the oracle tests and the
GPU engine tests.  They mirror how the PipeInfer driver uses the API (speculative.cpp:230-246 prompt tagged with every
seq id; :891-905 partition fork; :1017-1032 tree branch; :588-612 accept/prune; :1252-1268 non-spec run)."""
import numpy as np


def scenario_spec_like(n_vocab, seed=0, n_seq=4):
    rng = np.random.default_rng(seed)
    tok = lambda n: [int(x) for x in rng.integers(3, n_vocab, n)]
    all_seqs = list(range(n_seq))
    ops = []
    prompt = tok(8)
    # prompt minus last token as one batch, then the last token alone (speculative.cpp:235-246)
    ops.append(("decode", prompt[:7], list(range(7)), [all_seqs] * 7))
    ops.append(("decode", prompt[7:], [7], [all_seqs]))
    # non-speculative run for a sampled token at pos 8 (speculative.cpp:1252-1268)
    for i in all_seqs:
        ops.append(("seq_rm", i, 8, 9))
    ops.append(("decode", tok(1), [8], [[0]]))
    for i in all_seqs[1:]:
        ops.append(("seq_cp", 0, i, 8, 9))
    # speculative run in partition {1,2}: fork from canonical (:891-905), tree with one split (:1017-1032)
    for i in (1, 2):
        ops.append(("seq_rm", i, 9, -1))
        ops.append(("seq_cp", 0, i, 9, 10))
    d = tok(4)
    ops.append(("decode", d, [9, 10, 10, 11], [[1, 2], [1], [2], [1]]))
    # verify: branch 2 survived with 2 accepted tokens (positions 9,10) -> n_past = 11 (:588-612)
    for i in (1, 2):
        ops.append(("seq_rm", i, 11, -1))
    ops.append(("seq_rm", 0, 9, 11))
    ops.append(("seq_cp", 2, 0, 9, 11))
    for i in all_seqs[1:]:
        ops.append(("seq_rm", i, 9, 11))
        ops.append(("seq_cp", 0, i, 9, 11))
    # next non-spec token
    for i in all_seqs:
        ops.append(("seq_rm", i, 11, 12))
    ops.append(("decode", tok(1), [11], [[0]]))
    for i in all_seqs[1:]:
        ops.append(("seq_cp", 0, i, 11, 12))
    # a second speculative batch that reuses freed cells, 3 tokens, single branch
    ops.append(("seq_rm", 3, 12, -1))
    ops.append(("seq_cp", 0, 3, 12, 13))
    ops.append(("decode", tok(3), [12, 13, 14], [[3], [3], [3]]))
    # full rejection: drop the partition's tail
    ops.append(("seq_rm", 3, 12, -1))
    ops.append(("decode", tok(1), [12], [[0]]))
    return ops


def scenario_greedy(n_vocab, seed=1, n_prompt=8, n_gen=24):
    """plain target-only greedy continuation: the losslessness ground truth"""
    rng = np.random.default_rng(seed)
    return [int(x) for x in rng.integers(3, n_vocab, n_prompt)], n_gen


def run_scenario(ops, backend):
    """backend: object with decode(tokens,pos,seq_ids)->logits [n_tok,n_vocab], seq_rm, seq_cp, seq_keep.
    Returns the list of logits arrays, one per decode."""
    outs = []
    for op in ops:
        if op[0] == "decode":
            outs.append(np.array(backend.decode(op[1], op[2], op[3]), np.float32, copy=True))
        elif op[0] == "seq_rm":
            backend.seq_rm(*op[1:])
        elif op[0] == "seq_cp":
            backend.seq_cp(*op[1:])
        elif op[0] == "seq_keep":
            backend.seq_keep(*op[1:])
    return outs


def flatten_seq_ids(seq_ids, stride):
    n = len(seq_ids)
    flat = np.zeros((n, stride), np.int32)
    cnt = np.zeros(n, np.int32)
    for i, s in enumerate(seq_ids):
        cnt[i] = len(s)
        flat[i, :len(s)] = s
    return cnt, flat
