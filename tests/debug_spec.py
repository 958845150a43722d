import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import scenarios as S
from tests import gpu_util as U
from pipeinfer_amd import llama as LL
U.init()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
tgt = dft = "tiny_llama_q4_k_m"
g = np.load(os.path.join(GOLD, tgt + ".npz"))
prompt, _ = S.scenario_greedy(512, seed=11, n_prompt=8)
want = list(g["greedy_tokens"][:24])
for kw in (dict(n_draft=3, n_parallel=3, p_split=0.9), dict(n_draft=8, n_parallel=1, p_split=0.9), dict(n_draft=3, n_parallel=1, p_split=0.9, max_inflight=1)):
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
    out, res = LL.speculative(ct, cd, prompt, n_predict=23, p_accept=0.05, ignore_eos=0, **kw)
    print(kw); print(" out ", [int(x) for x in out]); print(" want", [int(x) for x in want]); print(" ", res)
