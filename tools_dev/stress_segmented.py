"""repeat the single-GPU arrangement (segmented target behind its feeder thread + priority draft) in ONE process until something breaks; not a test.
usage: PIPEINFER_CRASH_TRACE=1 python -X faulthandler tools_dev/stress_segmented.py [iterations]"""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pipeinfer_amd import llama as LL
from tests import gpu_util as U, scenarios as S
U.init()
GOLD = os.path.join(ROOT, "tests", "golden")
SEEDS = {"tiny_llama_b_q4_k_m": 13, "tiny_falcon40_q4_k_m": 15}
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.time()
for it in range(n_iter):
    for tgt, dft in (("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0")):
        g = np.load(os.path.join(GOLD, tgt + ".npz"))
        prompt, _ = S.scenario_greedy(512, seed=SEEDS[tgt], n_prompt=8)
        want = list(g["greedy_tokens"][:24])
        mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
        for min_spec in (0, 1, 3):
            ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
            ct.set_segments(1); cd.set_priority(1)
            if it % 2: ct.set_coalescing(8)
            out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=3, n_parallel=3, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=min_spec)
            assert out[:24] == want, (it, tgt, min_spec, out, want)
            ct.free(); cd.free()
        mt.free(); md.free()
    print(f"iteration {it} ok ({time.time() - t0:.0f} s)", flush=True)
print("stress: all iterations ok")
