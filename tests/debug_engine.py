import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import scenarios as S
from tests import gpu_util as U
from pipeinfer_amd import llama as LL
U.init()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEEDS = {"tiny_llama_q4_k_m": 11, "tiny_llama_q4_0": 12, "tiny_llama_b_q4_k_m": 13, "tiny_falcon_q4_0": 14, "tiny_falcon40_q4_k_m": 15}
for stem, seed in SEEDS.items():
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    m = LL.Model(os.path.join(GOLD, stem + ".gguf")); c = LL.Context(m, n_ctx=128, n_batch=64)
    class B:
        def decode(self, t, p, s): return c.decode(t, p, s)
        def seq_rm(self, *a): c.seq_rm(*a)
        def seq_cp(self, *a): c.seq_cp(*a)
        def seq_keep(self, *a): c.seq_keep(*a)
    outs = S.run_scenario(S.scenario_spec_like(m.n_vocab, seed=seed), B())
    errs = [float(np.abs(o - g[f"spec_like_{i}"]).max()) for i, o in enumerate(outs)]
    print(stem, "scale", float(np.abs(g["spec_like_0"]).max()), "errs", ["%.1e" % e for e in errs])
    c.kv_clear()
    prompt, n_gen = S.scenario_greedy(m.n_vocab, seed=seed)
    lg = c.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
    toks, cur = [], int(np.argmax(lg[-1]))
    for i in range(n_gen):
        toks.append(cur)
        lg = c.decode([cur], [len(prompt) + i], [[0]])
        srt = np.sort(lg[0])[::-1]
        if cur != int(g["greedy_tokens"][i]):
            print("  first mismatch at", i, "got", cur, "want", int(g["greedy_tokens"][i])); break
        cur = int(np.argmax(lg[0]))
        if i < 24 and srt[0] - srt[1] < 2e-3: print("  near tie at step", i, srt[0] - srt[1])
    c.free(); m.free()
