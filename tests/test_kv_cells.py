"""CPU: the engine's host KV-cell table (pipeinfer_amd/csrc/engine/kv_cells.cpp) against the oracle's restatement of the
reference (oracle/kvcache.c) AND against the reference's own functions (llama.cpp compiled into oracle/_ref/libref_kv.so) on random
PipeInfer-style operation sequences, and the closed-form range operations against
the loops of single operations they replace (speculative.cpp:603-612, 1252-1268)."""
import ctypes as C

import numpy as np
import pytest

from pipeinfer_amd import llama as LL
from tests import oracle_lib as O
from tests.scenarios import flatten_seq_ids

N_CTX, MAX_SEQ = 96, 40


class Eng:
    def __init__(self):
        self.lib = LL.api()
        self.h = self.lib.pi_kvcells_new(N_CTX)

    def decode(self, pos, seq_ids):
        cnt, flat = flatten_seq_ids(seq_ids, MAX_SEQ)
        p = np.asarray(pos, np.int32)
        return self.lib.pi_kvcells_decode(self.h, len(pos), O._ptr(p), O._ptr(cnt), O._ptr(flat), MAX_SEQ)

    def state(self):
        l = self.lib
        return (l.pi_kvcells_head(self.h), l.pi_kvcells_used(self.h), l.pi_kvcells_n_attend(self.h),
                [(l.pi_kvcells_pos(self.h, i), tuple(s for s in range(MAX_SEQ) if l.pi_kvcells_has(self.h, i, s))) for i in range(N_CTX)])

    def mask(self, n_kv, pos, seq_ids):
        cnt, flat = flatten_seq_ids(seq_ids, MAX_SEQ)
        p = np.asarray(pos, np.int32)
        out = np.zeros((len(pos), (n_kv + 31) // 32), np.uint32)
        self.lib.pi_kvcells_mask(self.h, n_kv, len(pos), O._ptr(p), O._ptr(cnt), O._ptr(flat), MAX_SEQ, O._ptr(out))
        bits = np.unpackbits(out.view(np.uint8).reshape(len(pos), -1), axis=1, bitorder="little")[:, :n_kv]
        return bits.astype(bool)


class Orc:
    def __init__(self):
        self.lib = O.oracle()
        self.h = self.lib.o_kv_new(N_CTX, MAX_SEQ)

    def decode(self, pos, seq_ids):
        cnt, flat = flatten_seq_ids(seq_ids, MAX_SEQ)
        p = np.asarray(pos, np.int32)
        self.lib.o_kv_pre_decode(self.h, len(pos))
        head = self.lib.o_kv_find_slot(self.h, len(pos), O._ptr(p), O._ptr(cnt), O._ptr(flat), MAX_SEQ)
        if head >= 0:
            self.lib.o_kv_post_decode(self.h, len(pos))
        return head

    def state(self):
        l = self.lib
        return (l.o_kv_head(self.h), l.o_kv_used(self.h), l.o_kv_n(self.h),
                [(l.o_kv_cell_pos(self.h, i), tuple(s for s in range(MAX_SEQ) if l.o_kv_cell_has(self.h, i, s))) for i in range(N_CTX)])

    def mask(self, n_kv, pos, seq_ids):
        cnt, flat = flatten_seq_ids(seq_ids, MAX_SEQ)
        p = np.asarray(pos, np.int32)
        out = np.zeros((len(pos), n_kv), np.float32)
        self.lib.o_kv_build_mask(self.h, n_kv, len(pos), O._ptr(p), O._ptr(cnt), O._ptr(flat), MAX_SEQ, O._ptr(out))
        return out == 0


class Ref:
    """the reference's own llama.cpp functions (oracle/_ref/libref_kv.so): find_slot, seq_rm / cp / keep, cell_max on the real llama_kv_cache.  The three
    decode-level lines around find_slot live inside llama_decode_internal, so they are applied here as the reference writes them: llama.cpp:5548-5550
    (head back to 0 when it ran far ahead of the used cells), :5571 (n = min(n_ctx, max(32, cell_max))), :5708-5713 (head advances past the slot)."""

    def __init__(self):
        self.lib = O.ref_kv()
        self.h = self.lib.refkv_new(N_CTX)

    def decode(self, pos, seq_ids):
        l, n = self.lib, len(pos)
        cnt, flat = flatten_seq_ids(seq_ids, MAX_SEQ)
        p = np.asarray(pos, np.int32)
        if l.refkv_head(self.h) > l.refkv_used(self.h) + 2 * n:
            l.refkv_set_head(self.h, 0)
        head = l.refkv_find_slot(self.h, n, O._ptr(p), O._ptr(cnt), O._ptr(flat), MAX_SEQ)
        if head >= 0:
            nh = head + n
            l.refkv_set_head(self.h, 0 if nh >= N_CTX else nh)
        return head

    def state(self):
        l = self.lib
        return (l.refkv_head(self.h), l.refkv_used(self.h), min(N_CTX, max(32, l.refkv_cell_max(self.h))),
                [(l.refkv_cell_pos(self.h, i), tuple(s for s in range(MAX_SEQ) if l.refkv_cell_has(self.h, i, s))) for i in range(N_CTX)])


def random_ops(rng, n_ops):
    ops = []
    n_past = 0
    for _ in range(n_ops):
        k = rng.integers(0, 10)
        if k < 4:
            n = int(rng.integers(1, 6))
            seqs = [sorted(set(int(x) for x in rng.integers(0, MAX_SEQ, rng.integers(1, 4)))) for _ in range(n)]
            if rng.random() < 0.2:
                seqs = [list(range(MAX_SEQ))] * n           # prompt-style: every sequence id
            pos = [n_past + i for i in range(n)]
            n_past += int(rng.integers(0, n + 1))
            ops.append(("decode", pos, seqs))
        elif k < 7:
            p0 = int(rng.integers(-1, max(n_past, 1) + 1)); p1 = int(rng.choice([-1, p0 + int(rng.integers(0, 4))]))
            ops.append(("rm", int(rng.integers(-1, MAX_SEQ)) if rng.random() < 0.9 else -1, p0, p1))
        elif k < 9:
            p0 = int(rng.integers(-1, max(n_past, 1) + 1)); p1 = int(rng.choice([-1, p0 + int(rng.integers(0, 4))]))
            ops.append(("cp", int(rng.integers(0, MAX_SEQ)), int(rng.integers(0, MAX_SEQ)), p0, p1))
        else:
            ops.append(("keep", int(rng.integers(0, MAX_SEQ))) if rng.random() < 0.3 else ("rm", 0, 0, 0))
    return ops


def apply(b, op):
    if op[0] == "decode":
        return b.decode(op[1], op[2])
    lib, h = b.lib, b.h
    pre = "pi_kvcells_" if isinstance(b, Eng) else "refkv_" if isinstance(b, Ref) else "o_kv_"
    if op[0] == "rm":
        getattr(lib, pre + "seq_rm")(h, *op[1:])
    elif op[0] == "cp":
        if op[1] != op[2]:                       # the public wrapper returns early for src == dst (llama.cpp:9303-9305)
            getattr(lib, pre + "seq_cp")(h, *op[1:])
    elif op[0] == "keep":
        getattr(lib, pre + "seq_keep")(h, op[1])


@pytest.mark.parametrize("seed", range(12))
def test_engine_cells_equal_oracle_cells(seed):
    rng = np.random.default_rng(seed)
    e, o = Eng(), Orc()
    for op in random_ops(rng, 120):
        assert apply(e, op) == apply(o, op), op
        assert e.state() == o.state(), op
        if op[0] == "decode":
            n_kv = e.state()[2]
            assert np.array_equal(e.mask(n_kv, op[1], op[2]), o.mask(n_kv, op[1], op[2]))


@pytest.mark.parametrize("seed", range(24))
def test_engine_and_oracle_cells_equal_the_reference_functions(seed):
    """pins the restatement (oracle/kvcache.c) AND the engine's bitset table to the reference's own code: the same random PipeInfer-style operation
    sequences through llama.cpp's llama_kv_cache_find_slot / seq_rm / seq_cp / seq_keep / cell_max, compiled from the reference sources"""
    if O.ref_kv() is None:
        pytest.skip("oracle/_ref/libref_kv.so is not built here (needs the reference sources)")
    rng = np.random.default_rng(1000 + seed)
    e, o, r = Eng(), Orc(), Ref()
    n_dec = 0
    for op in random_ops(rng, 200):
        got_r = apply(r, op)
        assert apply(e, op) == got_r and apply(o, op) == got_r, op
        sr = r.state()
        assert e.state() == sr and o.state() == sr, op
        if op[0] == "decode" and got_r >= 0:
            # the attention mask is written inside the graph callback (llama.cpp:5212-5232: cell i is hidden from token j unless it carries one of the token's
            # sequence ids and cells[i].pos <= pos[j]); stated here on the REFERENCE's cells
            n_kv, cells = sr[2], sr[3]
            want = np.array([[any(sq in cells[i][1] for sq in op[2][j]) and cells[i][0] <= op[1][j] for i in range(n_kv)] for j in range(len(op[1]))])
            assert np.array_equal(e.mask(n_kv, op[1], op[2]), want) and np.array_equal(o.mask(n_kv, op[1], op[2]), want), op
        n_dec += op[0] == "decode" and got_r is not None and got_r >= 0
    assert n_dec > 20


@pytest.mark.parametrize("seed", range(12))
def test_range_ops_equal_single_op_loops(seed):
    rng = np.random.default_rng(100 + seed)
    a, b = Eng(), Eng()
    lib = a.lib
    for op in random_ops(rng, 60):
        apply(a, op); apply(b, op)
        assert a.state() == b.state()
        first = int(rng.integers(0, MAX_SEQ)); last = int(rng.integers(first, MAX_SEQ))
        p0 = int(rng.integers(-1, 20)); p1 = int(rng.choice([-1, p0 + int(rng.integers(0, 6))]))
        kind = rng.integers(0, 3)
        if kind == 0:
            lib.pi_kvcells_seq_rm_range(a.h, first, last, p0, p1)
            for s in range(first, last + 1):
                lib.pi_kvcells_seq_rm(b.h, s, p0, p1)
        elif kind == 1:
            src = int(rng.integers(0, MAX_SEQ))
            if first == last == src:
                continue
            lib.pi_kvcells_seq_cp_range(a.h, src, first, last, p0, p1)
            for s in range(first, last + 1):
                if s != src:
                    lib.pi_kvcells_seq_cp(b.h, src, s, p0, p1)
        else:
            src = 0
            first = max(first, 1); last = max(last, first)
            lib.pi_kvcells_seq_rm_cp_range(a.h, src, first, last, p0, p1)
            for s in range(first, last + 1):
                lib.pi_kvcells_seq_rm(b.h, s, p0, p1)
                lib.pi_kvcells_seq_cp(b.h, src, s, p0, p1)
        assert a.state() == b.state(), (kind, first, last, p0, p1)


def test_hand_worked_placement():
    """first fit from head, wrap, head lowered by seq_rm to the first free cell, reset to 0 by seq_cp (llama.cpp:1593-1715)"""
    e = Eng()
    assert e.decode([0, 1, 2], [[0], [0], [0]]) == 0
    assert e.decode([3], [[0]]) == 3
    e.lib.pi_kvcells_seq_rm(e.h, 0, 1, 2)                      # frees cell 1; head (4) drops to 1
    assert e.state()[0] == 1 and e.state()[1] == 3
    assert e.decode([4, 5], [[0], [0]]) == 4                   # needs 2 contiguous cells: cell 1 alone does not fit
    assert e.decode([6], [[1]]) == 6
    e.lib.pi_kvcells_seq_cp(e.h, 0, 2, 0, -1)                  # head := 0
    assert e.state()[0] == 0
    assert e.decode([7], [[2]]) == 1                           # first free cell from the start
    st = e.state()
    assert st[3][1] == (7, (2,)) and st[3][0] == (0, (0, 2))
    m = e.mask(st[2], [8], [[2]])
    assert m[0, 0] and m[0, 1] and not m[0, 6]                 # seq 2 sees its own cells and the copied prefix, not seq 1's cell
