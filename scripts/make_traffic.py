"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh.

usage: make_traffic.py <prof_dir> > profiles/traffic.json
The stream probe in the same passes (lssvr_stream_probe over `probe_doubles` doubles: 8 B read +
8 B written per double) calibrates the two counters' units on this device; every kernel's HBM
bytes per launch = counter mean x calibration.  (MI355X_MICROARCH.md: separate --pmc passes,
KiB units, gfx950 FETCH_SIZE counts half the bytes -- the probe measures exactly that factor.)"""
import collections, csv, glob, json, sys

prof = sys.argv[1]
probe_doubles = int(sys.argv[2]) if len(sys.argv) > 2 else 12500000


def means(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def kernel_bytes(fetch_dir, write_dir, want, probe=None):
    out = {}
    for name, d, ctr in (("fetch", fetch_dir, "FETCH_SIZE"), ("write", write_dir, "WRITE_SIZE")):
        m = means(d)
        cal = None
        if probe is None:
            pv = [v for (k, c), v in m.items() if "stream_copy_probe" in k and c == ctr]
            cal = probe_doubles * 8.0 / (pv[0] * 1024.0) if pv else None
        else:
            cal = probe[name]
        wants = want if isinstance(want, (tuple, list)) else (want,)
        kv = [[v for (k, c), v in m.items() if w in k and c == ctr] for w in wants]
        out[name] = (sum(v[0] for v in kv) if all(kv) else None, cal)      # several kernels: their sum
    return out


res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (scripts/collect_profiles.sh), "
               "mean per dispatch in KiB, multiplied by the calibration the 8-byte-per-lane stream probe of the "
               "same passes gives (known bytes / counted KiB; scripts/make_traffic.py)"}
small = kernel_bytes(prof + "/fetch", prof + "/write", "enhance_small_kernel")
cal = {"fetch": small["fetch"][1], "write": small["write"][1]}
res["_calibration_bytes_per_counted_byte"] = cal
large = kernel_bytes(prof + "/fetchL", prof + "/writeL", ("moments_kernel", "solve4_parity_kernel"), probe=cal)
mfma = kernel_bytes(prof + "/fetchLm", prof + "/writeLm", "enhance_large_kernel", probe=cal)
shared = kernel_bytes(prof + "/fetchS", prof + "/writeS", "enhance_shared_kernel", probe=cal)
for key, kb, alg in (("M9_n16_ne100008", small, 88 * 100008), ("M33_n64_ne100000", large, 280 * 100000),
                     ("M33_n64_ne100000_mfma_kernel", mfma, 280 * 100000),
                     ("shared_M9_n16_ne10000000", shared, 88 * 10000000)):
    f, w = kb["fetch"][0], kb["write"][0]
    if f is None or w is None or cal["fetch"] is None:
        continue
    res[key] = {"fetch_size_kib": f, "write_size_kib": w,
                "hbm_bytes_per_launch": f * 1024 * cal["fetch"] + w * 1024 * cal["write"],
                "algorithmic_bytes_per_launch": alg}
print(json.dumps(res, indent=1))
