"""A conversation turn appended to a long cached prefix (prompt_cache continuation = LanguageModel._prefill_onto_cache): n new tokens
onto Tf cached ones, time per call - the prefix rows are keys only since round 6 (vlm_attn_prefill q_start)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from benchlib.common import _load_synthetic
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl

dev = torch.device("cuda", 0)
cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=65536, max_seqs=8)
lm = model.language_model
rng = np.random.default_rng(1)
for Tf in (2048, 8192, 16384):
    for n in (64, 512, 2048):
        best = 1e9
        for rep in range(3):
            c = lm.make_cache()
            emb = lambda k: lm._w["embed"][torch.from_numpy(rng.integers(1000, 100000, k)).to(dev)]      # noqa: E731
            pos = lambda a, b: np.broadcast_to(np.arange(a, b, dtype=np.int64)[None], (3, b - a)).copy()  # noqa: E731
            lm.prefill(emb(Tf), pos(0, Tf), [c], [Tf], "last", reserve_extra=n + 8)
            E2, P2 = emb(n), pos(Tf, Tf + n)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lm.prefill(E2, P2, [c], [n], "last", reserve_extra=8)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            c[0]._seq.release()
        print(f"prefix {Tf:6d} + {n:5d} new tokens: {best*1e3:8.2f} ms")
