/* libvihds_host.so -- host-side native code beside the HIP library (no device code, no HIP dependency).
 *
 * One job: numpy's LEGACY standard-normal stream, bit for bit, fast.  The reference draws u ~ N(0,1)[B,S,P] on the host every
 * step with np.random.randn (vihds/vae.py:22-24: 252 000 normals at B=36, S=200, P=35 = 2.3 ms of numpy, 30x the GPU's work for
 * the step).  A spec that keeps the reference's random stream (u_rng: numpy, the default) gets the SAME float32 numbers --
 * and the same global RandomState afterwards -- from here in 0.14 ms (8 threads), and the NEXT step's draw can run on a native
 * helper thread while Python queues the current step.  Algorithm and the bit-exactness argument: csrc/host/vihds_nprand.cpp.
 * Python binding: vi-hds_amd/vihds/nprand.py (ctypes).  Plumbing, not the hot path: when the library is missing the
 * binding falls back to numpy itself.
 */
#ifndef VIHDS_HOST_H
#define VIHDS_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIHDS_HOST_ABI_VERSION 4
int vihds_host_abi_version(void);
/* 1 when the CPU has the vector extensions the library was built for (AVX2); the binding uses numpy itself otherwise */
int vihds_host_cpu_ok(void);

/* out[0..n) <- np.random.standard_normal(n).astype(np.float32) of the RandomState (key[624], pos, has_gauss, gauss) -- numpy's
 * MT19937 state as np.random.get_state() returns it; the four are left as numpy would leave them.  n_threads: 1..64.
 * Returns 0, or -1 on bad arguments.  One call at a time (serialised inside). */
int vihds_np_randn_f32(uint32_t* key, int* pos, int* has_gauss, double* gauss, float* out, long long n, int n_threads);

/* The same draw on the library's helper thread.  start: even n, a state without a cached deviate; key / pos are read and
 * advanced IN PLACE (numpy's own state array: np.random.mtrand._rand._bit_generator.ctypes.state_address) -- nobody else may
 * use that generator until every started draw has been waited for.  Up to TWO draws may be started before the first is
 * waited for; the helper works through them in order.  Returns 0, -1 on bad arguments, -2 when two draws are already queued.
 * wait: blocks until the OLDEST draw started and not yet waited for is complete; returns its result (0), or -3 when there
 * is none. */
int vihds_np_randn_f32_start(uint32_t* key, int* pos, float* out, long long n, int n_threads);
int vihds_np_randn_f32_wait(void);

#ifdef __cplusplus
}
#endif
#endif
