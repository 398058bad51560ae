"""Golden vectors for the logits processors (repetition / presence / frequency penalties, logit_bias), produced by the
REFERENCE'S OWN `mlx_vlm/sample_utils.py::make_logits_processors` imported unmodified from /root/reference and executed
over oracle/mlx_shim (see make_golden_ref.py for the import machinery).  Run once in the build container:

    python tests/golden/make_golden_penalties.py        -> tests/golden/penalties_ref.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_ref as G  # noqa: E402


def main():
    mx, q, cfgm, cache, su = G.import_reference()
    assert su.__file__.startswith(G.REF)
    rng = np.random.default_rng(7)
    out = {}
    cases = [
        dict(logit_bias={5: 2.0, 7: -1.5}, rep=1.3, rep_ctx=20, pres=0.7, pres_ctx=20, freq=0.4, freq_ctx=20),
        dict(logit_bias=None, rep=1.1, rep_ctx=5, pres=None, pres_ctx=20, freq=None, freq_ctx=20),
        dict(logit_bias=None, rep=None, rep_ctx=20, pres=1.5, pres_ctx=3, freq=0.25, freq_ctx=64),
        dict(logit_bias={0: -100.0}, rep=0.8, rep_ctx=64, pres=0.0, pres_ctx=20, freq=1.0, freq_ctx=2),
    ]
    for ci, c in enumerate(cases):
        V = 257
        logits = (rng.standard_normal((2, V)) * 3).astype(np.float32)
        toks = rng.integers(0, 40, 70)                      # many repeats inside every window
        procs = su.make_logits_processors(c["logit_bias"], c["rep"], c["rep_ctx"], c["pres"], c["pres_ctx"], c["freq"],
                                          c["freq_ctx"])
        x = mx.array(logits).astype(mx.bfloat16)
        out[f"case{ci}.logits_bf16_as_f32"] = np.asarray(x.astype(mx.float32))
        for pfn in procs:
            x = pfn(mx.array(toks), x)
        out[f"case{ci}.tokens"] = toks
        out[f"case{ci}.out_bf16_as_f32"] = np.asarray(x.astype(mx.float32))
        for k, v in c.items():
            if k == "logit_bias":
                out[f"case{ci}.bias_idx"] = np.array(list((v or {}).keys()), dtype=np.int64)
                out[f"case{ci}.bias_val"] = np.array(list((v or {}).values()), dtype=np.float64)
            else:
                out[f"case{ci}.{k}"] = np.array(np.nan if v is None else v, dtype=np.float64)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "penalties_ref.npz"), **out)
    print("wrote penalties_ref.npz")


if __name__ == "__main__":
    main()
