"""Stage probe of int4_scan_tab_kernel (a -DVG_I4_TIMING build: tools/build_variant.sh i4t k_sq8.hip -DVG_I4_TIMING, run
with VECGO_HIP_LIB=variants/libvecgo_i4t.so): s_memtime cycles (100 MHz constant clock on gfx9: 10 ns per tick) per
wave by phase — waiting for the piece's global loads + issuing the staging writes, the staging round trip until the
first 16 code bytes are back, the 8 x 32 lookups of the piece.  argv: [N]."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import vecgo_amd as vg, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 768
ctx = vg.Context(0); dev = torch.device("cuda", 0)
rows = bench.gen_rows(0, 1_000_000, dev)
iq = vg.Int4Quantizer(ctx, D); iq.train(rows[:65536])
codes = iq.encode(rows).repeat((N + 999_999) // 1_000_000, 1)[:N].contiguous()
q = bench.gen_queries(1, dev)[0][0].contiguous()
out = torch.empty(N, device=dev)
for pre, fn in ((False, iq.l2_distance_batch), (True, iq.l2_distance)):
    for _ in range(100): fn(q, codes, out=out)
    torch.cuda.synchronize()
    waves = int(os.environ.get("I4_WAVES", "12"))
    t = out[:256 * waves * 8].cpu().numpy().reshape(-1, 8)
    tiles, total, vm, ld, look = (t[:, i].sum() for i in range(5))
    print(f"precomputed={pre}: {t.shape[0]} waves, {tiles / t.shape[0]:.1f} tiles per wave, ticks per tile: all {total / tiles:.1f} = "
          f"loads+writes {vm / tiles:.1f} + staging round trip {ld / tiles:.1f} + lookups {look / tiles:.1f} + rest {(total - vm - ld - look) / tiles:.1f}")
