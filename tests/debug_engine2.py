import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import scenarios as S
from tests import gpu_util as U, oracle_lib as O
from pipeinfer_amd import llama as LL
U.init()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
for stem, seed in {"tiny_llama_b_q4_k_m": 13, "tiny_falcon_q4_0": 14, "tiny_falcon40_q4_k_m": 15}.items():
    path = os.path.join(GOLD, stem + ".gguf")
    m = LL.Model(path); c = LL.Context(m, n_ctx=128, n_batch=64)
    om = O.OracleModel(path, n_ctx=128, max_seq=8)
    prompt, n_gen = S.scenario_greedy(m.n_vocab, seed=seed)
    for n_first in (len(prompt),):
        lg = c.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
        lo = om.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
        print(stem, "prompt err per token", ["%.1e" % e for e in np.abs(lg - lo).max(1)])
        cur = int(np.argmax(lo[-1]))
        for i in range(18):
            lg = c.decode([cur], [len(prompt) + i], [[0]])
            lo = om.decode([cur], [len(prompt) + i], [[0]])
            print("   step", i, "err %.1e" % np.abs(lg - lo).max(), "argmax", int(lg[0].argmax()), int(lo[0].argmax()))
            cur = int(np.argmax(lo[0]))
    c.free(); m.free(); om.close()
