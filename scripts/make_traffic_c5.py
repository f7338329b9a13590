"""Adds BASELINE config 5's keys to profiles/traffic.json from the passes of scripts/collect_c5.sh.

usage: make_traffic_c5.py <prof_dir> <traffic.json>   (rewrites the json in place)

FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950: a 128-byte (full-line) request is tallied at 64, a
64-byte (half-line) request at its 64.  Both calibrations are MEASURED in the same passes on known
byte counts: lssvr_stream_probe (full lines: x ~2.0) and lssvr_row_chunk_probe with chunks of 8
(the half-line pattern of the element-major staging: x 1.0) and of 16 (full lines again).
  point-major kernel  : every read is a full-line request -> FETCH x cal_stream
  element-major kernel: the 384 B of tabulated rows per element are half-line requests (x cal_rows8),
                        the 16 B of x, u full-line (x cal_stream)."""
import collections, csv, glob, json, sys

prof, out = sys.argv[1], sys.argv[2]
NE, N, M = 1000008, 16, 9
PROBE = 12500000


def means(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(m, frag, ctr):
    v = [val for (k, c), val in m.items() if frag in k and c == ctr]
    return v[0] if v else None


tj = json.load(open(out))
res = {}
for tag, fd, wd, kern in (("pm", "fetch", "write", "enhance_small_kernel<9, 2, true"),
                          ("em", "fetch_em", "write_em", "enhance_small_kernel<9, 0, true")):
    mf, mw = means(prof + "/" + fd), means(prof + "/" + wd)
    cal_stream = PROBE * 8.0 / (pick(mf, "stream_copy_probe", "FETCH_SIZE") * 1024.0)
    cal_rows8 = NE * N * 8.0 / (pick(mf, "row_chunk_probe_kernel<8>", "FETCH_SIZE") * 1024.0)
    cal_rows16 = NE * N * 8.0 / (pick(mf, "row_chunk_probe_kernel<16>", "FETCH_SIZE") * 1024.0)
    cal_w = PROBE * 8.0 / (pick(mw, "stream_copy_probe", "WRITE_SIZE") * 1024.0)
    f_kib, w_kib = pick(mf, kern, "FETCH_SIZE"), pick(mw, kern, "WRITE_SIZE")
    if tag == "pm":
        read = f_kib * 1024 * cal_stream
    else:
        xu_counted = 16.0 * NE / cal_stream                       # full-line reads of x, u as the counter sees them
        read = (f_kib * 1024 - xu_counted) * cal_rows8 + 16.0 * NE
    res[tag] = {"fetch_size_kib": f_kib, "write_size_kib": w_kib,
                "calibration_bytes_per_counted_byte": {"full_line_stream": cal_stream, "half_line_rows_of_8": cal_rows8,
                                                       "full_line_rows_of_16": cal_rows16, "write": cal_w},
                "hbm_bytes_per_launch": read + w_kib * 1024 * cal_w,
                "algorithmic_bytes_per_launch": 472 * NE,
                "note": ("point-major tables: every request a full line" if tag == "pm" else
                         "element-major tables: 64-byte half-line requests for the tabulated rows (counted at face value), "
                         "full lines for x, u")}
tj["c5_M%d_n%d_ne%d" % (M, N, NE)] = res["pm"]
tj["c5_element_major_M%d_n%d_ne%d" % (M, N, NE)] = res["em"]
tj["_round_c5"] = "round 3 (scripts/collect_c5.sh)"
json.dump(tj, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
