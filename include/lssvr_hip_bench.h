/* Measurement entries of liblssvr_hip.so (same library as include/lssvr_hip.h; NOT part of the product ABI).
 *
 * What bench.py, scripts/ and the rocprofv3 collection scripts use to time single launches and to calibrate the
 * counters: blocking launches stamped with the dispatch's own begin / end times, sequences of stamped launches,
 * and three microbenchmark probes.  A caller of the hot path needs none of them.  Conventions (device pointers,
 * return codes, lssvr_last_error) are those of lssvr_hip.h.
 */
#ifndef LSSVR_HIP_BENCH_H
#define LSSVR_HIP_BENCH_H

#include "lssvr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * lssvr_enhance_ws_sequence -- `repeats` launches of lssvr_enhance_ws back to back on `stream`, each
 * stamped with its own begin / end timestamps (what rocprofv3 --kernel-trace reports per dispatch), ONE
 * synchronisation at the end: the duration of the launch INSIDE a running sequence, where
 * lssvr_enhance_profiled measures it in isolation (an idle chip before and after).  BLOCKING measurement
 * aid for bench.py's roofline.  kernel_ms_host: float[repeats] on the host.
 */
int lssvr_enhance_ws_sequence(const double* x, const double* u, int64_t ne,
                              int64_t elem_offset, int64_t ne_global,
                              double gxmin, double gxmax, double bc_left, double bc_right,
                              int M, int n_colloc, double gamma,
                              int rhs_id, const double* rhs_params_host, const double* rhs_values,
                              int solver_id,
                              double* W, int32_t* status, int32_t* fail_count,
                              void* work, int64_t work_bytes, void* stream,
                              int repeats, float* kernel_ms_host);

/* The same for lssvr_enhance_varcoef_ws (BASELINE config 5). */
int lssvr_enhance_varcoef_ws_sequence(const double* x, const double* u, int64_t ne,
                                      int64_t elem_offset, int64_t ne_global,
                                      double gxmin, double gxmax, double bc_left, double bc_right,
                                      int M, int n_colloc, double gamma,
                                      const double* a_values, const double* da_values,
                                      const double* rhs_values, int table_layout,
                                      double* W, int32_t* status, int32_t* fail_count,
                                      void* work, int64_t work_bytes, void* stream,
                                      int repeats, float* kernel_ms_host);

/*
 * lssvr_enhance_profiled -- the same launch as lssvr_enhance, stamped with the
 * dispatch's own begin/end timestamps (hipExtLaunchKernelGGL).  BLOCKING: waits for
 * the kernel and returns its duration in *kernel_ms_host.  Measurement aid for
 * bench.py's roofline; not for production pipelines.
 */
int lssvr_enhance_profiled(const double* x, const double* u, int64_t ne,
                           int64_t elem_offset, int64_t ne_global,
                           double gxmin, double gxmax, double bc_left, double bc_right,
                           int M, int n_colloc, double gamma,
                           int rhs_id, const double* rhs_params_host, const double* rhs_values,
                           int solver_id,
                           double* W, int32_t* status, void* stream, float* kernel_ms_host);

/*
 * lssvr_fp64_probe -- FP64 FMA throughput microbenchmark used to quote the
 * roofline peak: runs `iters` dependent-free fused multiply-adds per lane on
 * `blocks` x 256 threads; out[blocks*256] receives a checksum.  flops = 2 * 8 *
 * iters * blocks * 256 (8 independent accumulators per lane).  use_mfma: 0 = v_fma_f64,
 * 1 = v_mfma_f64_16x16x4_f64, 3 = v_mfma_f64_4x4x4_4b_f64 (8 per iteration), 2 = MFMA and FMA
 * workgroups interleaved (do the two pipes overlap? they do not), >= 100 = FMA with
 * (use_mfma - 100) active lanes per wave.
 */
int lssvr_fp64_probe(double* out, int blocks, int iters, int use_mfma, void* stream);

/*
 * lssvr_stream_probe -- dst[i] = src[i] + 1 over n doubles with 8-byte-per-lane
 * accesses (8n bytes read, 8n written): a known byte count in the enhancement
 * kernels' access width, used to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE.
 */
int lssvr_stream_probe(const double* src, double* dst, int64_t n, void* stream);

/*
 * lssvr_row_chunk_probe -- dst[r] = sum of row r of src[nrows*rowlen], read the way the lane
 * kernels stage tabulated inputs: a wave owns 64 rows and reads `chunk` (8 or 16) columns of them per
 * batch, consecutive lanes on consecutive doubles (chunk*8-byte runs, rowlen*8 bytes apart).
 * nrows*rowlen*8 bytes read, nrows*8 written: calibrates FETCH_SIZE for that pattern (chunk = 8:
 * half-line requests) and measures the bandwidth the pattern reaches (BASELINE config 5's a, a', f rows).
 */
int lssvr_row_chunk_probe(const double* src, double* dst, int64_t nrows, int rowlen, int chunk, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSSVR_HIP_BENCH_H */
