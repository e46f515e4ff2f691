import sys; sys.path.insert(0, '/root/repo')
import numpy as np, subprocess, os
if len(sys.argv) > 2:
    import vecgo_amd as vg
    from tests import hooks
    d = np.load(sys.argv[1], allow_pickle=True)
    cfg = eval(str(d["cfg"])); x, xr, q = d["x"], d["xr"], d["q"]
    mode = sys.argv[2]
    ctx = vg.Context(0)
    dim = cfg["dim"]
    if mode == "clean": xr = x
    if mode == "nan_only": xr = np.where(np.isnan(xr), xr, x)
    if mode == "inf_only": xr = np.where(np.isinf(xr), xr, x)
    if mode == "big_only": xr = np.where(np.isfinite(xr) & (np.abs(xr) > 1e20), xr, x)
    if mode == "dim64":
        xr = np.concatenate([xr] * 4, axis=1); q = np.concatenate([q] * 4, axis=1); dim = 64
    if mode == "metric0": cfg["metric"] = 0
    print("non-finite rows:", np.flatnonzero(~np.isfinite(xr).all(1))[:10], "huge:", np.flatnonzero((np.abs(np.nan_to_num(xr)) > 1e20).any(1))[:10], flush=True)
    idx = vg.Index(ctx, cfg["n"], dim, vg.Metric(cfg["metric"])); idx.set_vectors(xr)
    idx.enable_bf16_filter(True)
    ids, sc = idx.search_flat(q, min(cfg["k"], 512))
    print("ok", ids[0][:5])
else:
    for mode in ("poisoned", "clean", "nan_only", "inf_only", "big_only", "dim64", "metric0"):
        r = subprocess.run([sys.executable, __file__, sys.argv[1], mode], capture_output=True, text=True)
        out = (r.stdout + r.stderr).strip().splitlines()
        print(mode, "->", r.returncode, [l[:120] for l in out if "amdgpu.ids" not in l][-2:], flush=True)
