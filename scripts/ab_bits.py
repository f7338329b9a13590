"""Bitwise A/B of two builds of the library on the same inputs: one process per library (LSSVR_HIP_LIB), W and status
of a few launches hashed and compared.  usage: ab_bits.py libA.so libB.so -- ne,M,n,x0,h [...]   (x0, h: mesh start / width)"""
import hashlib, json, os, subprocess, sys
if "--child" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from hybrid_fem_lssvr_amd import ops
    out = {}
    for spec in sys.argv[sys.argv.index("--child") + 1:]:
        ne, M, n, x0, h = spec.split(",")
        ne, M, n, x0, h = int(ne), int(M), int(n), float(x0), float(h)
        nodes = x0 + h * np.arange(ne + 1)
        u = np.sin(np.pi * nodes)
        W, st = ops.enhance(torch.as_tensor(nodes, device="cuda:0"), torch.as_tensor(u, device="cuda:0"), M, 1e4, n,
                            global_domain=(nodes[0], nodes[-1]))
        torch.cuda.synchronize()
        out[spec] = [hashlib.sha256(W.cpu().numpy().tobytes()).hexdigest()[:16], int(st.sum()), float(W.abs().max())]
    print(json.dumps(out))
    sys.exit(0)
sep = sys.argv.index("--")
libs, specs = sys.argv[1:sep], sys.argv[sep + 1:]
res = []
for lib in libs:
    env = dict(os.environ, LSSVR_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, __file__, "--child"] + specs, env=env, capture_output=True, text=True)
    try:
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    except Exception:
        print("FAILED", lib, r.stdout[-300:], r.stderr[-600:])
        sys.exit(1)
bad = 0
for s in specs:
    same = all(r[s] == res[0][s] for r in res)
    bad += not same
    print(s, "bit-equal" if same else "DIFFERENT", [r[s] for r in res])
sys.exit(1 if bad else 0)
