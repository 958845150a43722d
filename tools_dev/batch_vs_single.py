"""GPU debug: logits of a k-token batch must equal the logits of the same tokens decoded one by one (same cells)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from pipeinfer_amd import llama as LL, lib as L
import bench as B
lib = L.load(); L.check(lib.pi_device_init(0))
wl = sys.argv[1] if len(sys.argv) > 1 else "7b"
tspec, tft, _, _ = B.WORKLOADS[wl]
path = B.ensure_model("/tmp/pipeinfer_models", tspec, tft, 0x5EED, 1.0, 1.0)
m = LL.Model(path)
rng = np.random.default_rng(5)
prompt = [int(x) for x in rng.integers(3, m.n_vocab, 32)]
extra = [int(x) for x in rng.integers(3, m.n_vocab, 8)]
for k in (2, 3, 4, 8):
    a, b = LL.Context(m, n_ctx=1024, n_batch=512), LL.Context(m, n_ctx=1024, n_batch=512)
    a.decode(prompt, list(range(32)), [[0]] * 32); b.decode(prompt, list(range(32)), [[0]] * 32)
    singles = [a.decode([extra[i]], [32 + i], [[0]])[0] for i in range(k)]
    batch = b.decode(extra[:k], [32 + i for i in range(k)], [[0]] * k)
    for i in range(k):
        same = np.array_equal(singles[i].view(np.uint32), batch[i].view(np.uint32))
        print(f"k={k} token {i}: {'same' if same else 'DIFF max abs %.3e argmax %d vs %d' % (np.abs(singles[i] - batch[i]).max(), singles[i].argmax(), batch[i].argmax())}")
    a.free(); b.free()
