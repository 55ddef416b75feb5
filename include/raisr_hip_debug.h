/* raisr_hip_debug.h -- introspection hooks of the test suite and of bench.py's self-check legs.
 *
 * NOT part of the product ABI: libraisr_hip.so does not export these.  They exist in builds of the same sources with
 * -DRAISR_HIP_TESTHOOKS (`make libraisr_hip_testhooks.so`; development builds, -DRAISR_HIP_DEV, imply it), together with the
 * pipelines that are kept for comparisons only (RAISR_HIP_SPLIT=1: certified hash stage and filter stage as separate launches;
 * RAISR_HIP_DEFER=1: the exact path of the uncertified pixels as a separate kernel).  The reference has no counterpart of any of them. */
#ifndef RAISR_HIP_DEBUG_H
#define RAISR_HIP_DEBUG_H
#include "raisr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Introspection for tests / profiling ---------------------------------------------------------
 * Copies the last frame's per-pixel hash plane (u8: bucket 0..215 of the first hash, stale outside the
 * filtered zone) and HR plane (fp32; binary16 bit patterns in the low half-words in FP16 mode) of pass
 * `pass_index` to host buffers (either may be NULL).
 * The hash plane is only materialised for frames processed after raisr_hip_debug_keep_stages(ctx, 1): the
 * production kernel keeps the hashes on chip. */
int raisr_hip_debug_keep_stages(raisr_hip_ctx *ctx, int on);
int raisr_hip_debug_read_stage(raisr_hip_ctx *ctx, int pass_index, uint8_t *hash_out, float *hr_out);
/* Certified hash stage (the production kernel of the fp32 numerics computes the structure tensor approximately and
 * sends only the pixels whose bucket it cannot certify through the reference's exact instruction sequence).
 *   collect != 0: count, over the frames processed from now on, {pixels sent to the exact path, certified buckets that
 *                 differed from the exact ones (only counted with check != 0; must stay 0), filtered pixels};
 *   check   != 0: self-check mode -- EVERY pixel also takes the exact path and certified buckets are compared with it.
 * raisr_hip_debug_certify_stats() synchronises the context's stream and reads the counters: out[0..2] as above, then the per-tile view
 * of the worklist -- out[3] tiles (64 x 16 pixels) with a non-empty list, out[4] tiles whose list overflowed (they pay the approximate
 * AND the all-exact stage), out[5] tiles processed, out[6] flat tiles (hash stage skipped), out[7] reserved; read out[3..6] from a run
 * without the self-check, which lists every pixel.
 * Replaces nothing in the reference; it is the observability of an optimisation the reference does not have. */
int raisr_hip_debug_certify(raisr_hip_ctx *ctx, int collect, int check);
int raisr_hip_debug_certify_stats(raisr_hip_ctx *ctx, unsigned out[8]);
/* Multi-device ring, model hand-over (raisr_hip_stream_set_model with more than one device slot): number of per-device staging blobs
 * currently allocated.  With RAISR_HIP_TEST_FAIL_BLOB_SLOT=k in the environment the allocation on slot k fails (fault injection): the
 * call must return RAISR_HIP_ENOMEM and leave this count at 0 -- nothing of slots 0..k-1 stays behind. */
int raisr_hip_debug_stream_live_blobs(void);
/* Class-1 sign table of the certified hash stage (exactly one-dimensional windows, docs/CERTIFY.md s9): 65 536 bytes, entry i = the floats
 * whose mantissa >> 7 is i; bit 0: some a of them has fl(a/2 - VRCP14(VRSQRT14(fl(a a)/4))) < 0, bit 1: some has it >= 0.
 * RAISR_HIP_ESTATE when the class is switched off (RAISR_HIP_C1=0 at context creation). */
int raisr_hip_debug_read_c1tab(raisr_hip_ctx *ctx, uint8_t *out65536);
/* Decision of the certified hash stage for `n` host-side APPROXIMATE tensor triples (a', b', d'): the bucket it computes and
 * whether it certifies it (1) or would send the pixel to the exact path (0), by the device function the kernels run.
 * *eps_out (optional) receives the relative tensor error eps the certification assumes: a certified bucket must equal the
 * reference hash of EVERY exact tensor with |a-a'| <= eps a', |d-d'| <= eps d', |b-b'| <= eps (a'+d')/2
 * (tests/test_gpu_certify.py samples that box).  Needs a configured context (the weights depend on the bit depth). */
int raisr_hip_debug_approx_hash(raisr_hip_ctx *ctx, int pass_index, int hash_flavour, const float *abd, size_t n,
                                uint8_t *bucket_out, uint8_t *cert_out, float *eps_out);
/* Exhaustive self-check of the binary16 hash's folded thresholds ("Binary16 pipeline"): every operand pair the
 * fast hash can see goes through the two divisions the reference has (VDIVPH) and through the comparisons that replace them;
 * out[0] = disagreements (0 expected), out[1] = pairs compared.  Needs the pass's model. */
int raisr_hip_debug_fold16_check(raisr_hip_ctx *ctx, int pass_index, unsigned long long out[2]);

/* Hash bucket (0..215) of `n` host-side structure-tensor triples (a, b, d) x n with pass `pass_index`'s
 * thresholds, computed by the very device functions the hash kernel runs (fast path plus generic fall-back
 * for RAISR_HIP_HASH_AVX512; the RCPPS/RSQRTPS flavour for RAISR_HIP_HASH_AVX2).  Replaces nothing in the
 * reference: it exposes GetHashValue_AVX512_32f_16Elements (Library/Raisr_AVX512.cpp:175-258) and
 * GetHashValue_AVX256_32f_8Elements (Library/Raisr_AVX256.cpp:393-472) to unit tests on arbitrary inputs. */
int raisr_hip_debug_hash(raisr_hip_ctx *ctx, int pass_index, int hash_flavour, const float *abd, size_t n,
                         uint8_t *hash_out);
/* Times `iters` launches of each kernel of the configured pipeline on device-resident scratch
 * input with HIP events on the context's stream; writes per-kernel average milliseconds.
 * names_out receives up to max_kernels NUL-terminated names (64 bytes each). */
int raisr_hip_profile_kernels(raisr_hip_ctx *ctx, const void *d_in, size_t in_pitch, void *d_out,
                              size_t out_pitch, int iters, char *names_out, float *ms_out, int max_kernels);

/* development builds (-DRAISR_HIP_DEV) only: wave cycles per phase of the fused kernel since the last call (scripts/phase_cycles.py) */
int raisr_hip_dev_phase_stats(unsigned long long out[8]);

#ifdef __cplusplus
}
#endif
#endif /* RAISR_HIP_DEBUG_H */
