"""A/B timing of kernel variants: one process per library (LSSVR_HIP_LIB), hipExt-stamped
kernel durations, interleaved rounds.  usage: ab_kernel.py lib1.so lib2.so ... -- ne,M,n [...]"""
import os, subprocess, sys, json
if "--child" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from hybrid_fem_lssvr_amd import ops
    cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[sys.argv.index("--child") + 1:]]
    out = {}
    for ne, M, n in cfgs:
        x = torch.linspace(-1, 1, ne + 1, dtype=torch.float64, device="cuda:0")
        u = torch.sin(np.pi * x)
        W = torch.empty((ne, M), dtype=torch.float64, device="cuda:0")
        ts = sorted(ops.enhance_profiled(x, u, M, 1e4, n, global_domain=(-1.0, 1.0), out=W) for _ in range(60))
        out[f"{ne},{M},{n}"] = (ts[len(ts) // 2] * 1e6, ts[0] * 1e6)
    print(json.dumps(out))
    sys.exit(0)
sep = sys.argv.index("--")
libs, cfgs = sys.argv[1:sep], sys.argv[sep + 1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, LSSVR_HIP_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"] + cfgs, env=env, capture_output=True, text=True)
        try:
            res = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"round {rnd} {os.path.basename(lib):28s} " +
                  "  ".join(f"[{k}] med {v[0]:8.2f} us min {v[1]:8.2f}" for k, v in res.items()), flush=True)
        except Exception:
            print("FAILED", lib, r.stdout[-300:], r.stderr[-300:])
