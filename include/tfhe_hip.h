/*
 * tfhe_hip.h -- C ABI of the MI355X (gfx950) gate-bootstrapping engine.
 *
 * This is the drop-in boundary for go-tfhe's hot path.  The reference is pure Go
 * with no FFI layer; each entry point below replaces one of its in-process seams
 * (file:line relative to the go-tfhe tree) and is what a cgo shim binds
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative
 *     TFHE_E_* code and never throws; tfhe_last_error() gives the message.  The
 *     reference panics on every error on this path; the cgo shim turns rc != 0 into
 *     panic() to keep that behaviour.
 *   - a context belongs to ONE GPU.  Its calls are thread-safe and take effect one after the other, like calls on one
 *     evaluator.Evaluator (evaluator.go:14-24) -- except tfhe_gate_batch, whose concurrent callers are COMBINED into
 *     one launch (see there): many threads issuing scalar gates on one context is a supported, fast pattern.
 *   - keys are copied at load time and are immutable afterwards.
 *   - outputs are caller-owned (the *Assign style of the reference).
 *   - "_dev" variants take DEVICE pointers and a hipStream_t (as void*; NULL = HIP's
 *     default stream, as for any hipStream_t) and ONLY ENQUEUE work on that stream, ordered
 *     with the caller's other work there: no device memory is read back, nothing is
 *     synchronised, so a sequence of them can be captured into a hipGraph (size the context's
 *     intermediate buffers first: tfhe_ctx_reserve, or one un-captured call of the same batch
 *     size -- hipMalloc is not allowed during capture).  The others take HOST pointers, run on
 *     the context's private stream and return after the result is in the output buffer.
 *     All calls of one context are serialised by a mutex while they reserve and enqueue, but
 *     the "_dev" calls of one context share its intermediate device buffers: their work must be
 *     ordered on ONE stream at a time (stream order keeps it correct), and "_dev" work must
 *     not be in flight while a host-pointer call of the same context runs.  For independent
 *     streams use one context per stream.
 *   - all ciphertext words are uint32 torus values (params.Torus, params.go:27);
 *     an LWE sample is n+1 words with the body LAST (tlwe.go:11-33); a TRLWE sample is
 *     [2][N] words, A then B (trlwe.go:13-16).
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFHE_OK            0
#define TFHE_E_INVALID    -1   /* bad argument / unsupported parameter set          */
#define TFHE_E_NOKEY      -2   /* bootstrapping or key-switching key not loaded    */
#define TFHE_E_HIP        -3   /* HIP runtime error (message has the hipError_t)   */
#define TFHE_E_NOMEM      -4

/* The fields of params.TRGSWLv1Params / TLWELv0Params the path reads
 * (params.go:60-78).  Replaces the process-global params.CurrentSecurityLevel
 * (params.go:47): parameters are explicit per context. */
typedef struct {
    int32_t n;        /* TLWELv0.N                       */
    int32_t N;        /* TRGSWLv1.N  (512, 1024 or 2048) */
    int32_t Nbit;     /* TRGSWLv1.NBIT                   */
    int32_t L;        /* TRGSWLv1.L                      */
    int32_t Bgbit;    /* TRGSWLv1.BGBIT                  */
    int32_t basebit;  /* TRGSWLv1.BASEBIT                */
    int32_t t;        /* TRGSWLv1.IKS_T                  */
} tfhe_params;

/* Gate op codes for tfhe_gate_batch*: gates.go:26-114 / gates_helper.go:10-63. */
enum {
    TFHE_OP_NAND = 0, TFHE_OP_AND = 1, TFHE_OP_OR = 2, TFHE_OP_XOR = 3, TFHE_OP_XNOR = 4,
    TFHE_OP_NOR = 5, TFHE_OP_ANDNY = 6, TFHE_OP_ANDYN = 7, TFHE_OP_ORNY = 8, TFHE_OP_ORYN = 9,
    TFHE_OP_MUX = 10
};

typedef struct tfhe_ctx tfhe_ctx;

const char *tfhe_last_error(void);
/* "release" for every build that may be used; "control:fuzz" / "control:tsan" for the two test-only variants that misbehave on purpose
 * (the differential fuzzer's and the ThreadSanitizer script's positive controls).  A loader should refuse anything but "release". */
const char *tfhe_build_flavor(void);
/* Number of visible GPUs (hipGetDeviceCount). */
int tfhe_device_count(int *count);

/* evaluator.NewEvaluator (evaluator.go:27-35) + the CloudKey fields that are pure
 * functions of the parameters (cloudkey.go:60-85: decomposition offset, gate test
 * vector).  Supported parameter shapes: N=1024 with (L,Bgbit) = (3,6) [80/110/128-bit],
 * (2,10) [Uint1], (1,23) [Uint3]; N=2048 with (1,22) [Uint4, Uint5, and the shapes of
 * Uint6-8]; N=512 with (1,18) [Uint2]; any n < 1280, any key-switch (basebit, t) with
 * N*t <= 9216.  Anything else returns TFHE_E_INVALID. */
int tfhe_ctx_create(const tfhe_params *params, int device_id, tfhe_ctx **out);
int tfhe_ctx_destroy(tfhe_ctx *ctx);
int tfhe_ctx_params(const tfhe_ctx *ctx, tfhe_params *out);
/* Waits for the context's private stream and for everything "_dev" calls of this context have enqueued so far on
 * ANY caller stream (the context records an event of its own behind each such call -- it never touches a caller's
 * stream handle again, so streams may be destroyed freely), then reports (TFHE_E_INVALID) and clears anything the
 * kernels recorded since the last call: an op code outside TFHE_OP_NAND..TFHE_OP_MUX, or a MUX item without a third
 * operand, in a tfhe_gate_batch_dev whose op codes the host never sees.  Such items ran as plain bootstraps of their
 * first operand.  tfhe_ctx_destroy waits for the same events before it frees anything. */
int tfhe_ctx_sync(tfhe_ctx *ctx);
/* Sizes the context's intermediate device buffers for "_dev" batches of up to max_batch items (with_mux: for gate
 * batches that may contain MUX items) so that later calls allocate nothing -- required before capturing "_dev"
 * calls into a hipGraph.  Work is issued in slabs of 64 co-resident launches (65,536 bootstraps at N = 1024), so the
 * buffers stop growing there: 0.54 GB, 1.07 GB with MUX.
 * Growth rules: the buffers are grow-only and growing re-allocates.  A "_dev" call that would have to grow one while
 * its stream is being captured returns TFHE_E_INVALID (message names this function) instead of a HIP error; and once
 * any "_dev" call HAS been captured the context is FROZEN -- the graph holds the buffers' addresses -- so every later
 * call of any kind that needs larger buffers returns TFHE_E_INVALID rather than freeing memory a graph replay would
 * touch.  Clear TFHE_OPT_FROZEN after destroying the graphs to allow growth again. */
int tfhe_ctx_reserve(tfhe_ctx *ctx, int max_batch, int with_mux);
/* The same for tfhe_bootstrap_extended_batch_dev with polyExtendFactor ext (N = 2048 shape): the factors other than 2 keep
 * two sets of ext accumulators and the mod-switched samples of one chunk of the batch, which tfhe_ctx_reserve does not size. */
int tfhe_ctx_reserve_extended(tfhe_ctx *ctx, int max_batch, int ext);

/* Per-context options.  Kernel dispatch is a function of the parameters and the batch size only; these exist for
 * measurements (tools/) and tests, and are deliberately NOT read from the environment: a service must not change
 * kernels because of a stray variable.  value < 0 restores the default of the three limits. */
enum {
    TFHE_OPT_QUAD_MAX = 1,     /* N = 1024: launches of up to this many bootstraps use the small-batch (four-/eight-wave)
                                  kernels; default = number of CUs; 0 = always the two-wave kernel                       */
    TFHE_OPT_OCT_MAX = 2,      /* ... and of those, up to this many the eight-wave kernel; default = number of CUs       */
    TFHE_OPT_KS_MFMA_MIN = 3,  /* batches of at least this many ciphertexts use the matrix-core key switch where the key
                                  shape has one (default 24: below, the per-ciphertext gather is quicker); 0 = never
                                  (the vector-ALU kernels).  Every key-switch kernel is bit-exact: same words either way  */
    TFHE_OPT_FROZEN = 4,       /* 1 while a captured hipGraph may hold the intermediate buffers' addresses (set by the
                                  library, see tfhe_ctx_reserve); the caller clears it once those graphs are destroyed    */
    TFHE_OPT_COMBINE_MAX = 5,  /* tfhe_gate_batch calls of at most this many gates are COMBINED with concurrent callers'
                                  (see tfhe_gate_batch); default (and -1) = one per CU (also the most a combined launch
                                  carries: the rows of its page-locked staging), 0 = never                                  */
    TFHE_OPT_COMBINE_LAUNCHES = 6,  /* read-only: combined launches issued so far ...                                       */
    TFHE_OPT_COMBINE_REQUESTS = 7,  /* ... and the tfhe_gate_batch calls they carried                                       */
    TFHE_OPT_COMBINE_US_IDLE = 11,  /* read-only, microseconds summed over the combined launches of back-to-back rounds: from the previous
                                       launch's completion to this one's issue (the GPU-idle gap combining leaves between rounds) ...    */
    TFHE_OPT_COMBINE_US_GATHER = 12,/* ... the part of it the launch's leader spent waiting for the previous launch's callers to return ... */
    TFHE_OPT_COMBINE_US_LAUNCH = 13,/* ... and transfers + kernels + synchronisation of the combined launches themselves                   */
    TFHE_OPT_COMBINE_QUIET_US = 14, /* how long (microseconds, 20 ... 5000) the leader of a combined launch waits WITHOUT a new arrival for the callers of
                                       the previous launch before it goes without them (four such windows at most); -1 = default                    */
    TFHE_OPT_KS_WIDE_CT = 8,   /* wide key switch (bases 16-64): ciphertexts per wave, 64 (default: 0, -1) or 128 (measurements
                                  only: one wave per SIMD, slower)                                                         */
    TFHE_OPT_CLONE_PATH = 9    /* read-only: how tfhe_ctx_clone_to brought this context's keys here: 0 = not a clone, 1 = same
                                  GPU (device-to-device copy), 2 = peer copy GPU to GPU (xGMI), 3 = staged through page-locked
                                  host memory (the devices are not peers)                                                  */
};
int tfhe_ctx_set_option(tfhe_ctx *ctx, int option, int value);
int tfhe_ctx_get_option(tfhe_ctx *ctx, int option, int *value);

/* CloudKey.BootstrappingKey (cloudkey.go:16-21, []*trgsw.TRGSWLv1FFT, trgsw.go:60-68),
 * flattened by the shim to [n][2L][2][N] float64: row r < L multiplies the digits of A,
 * row L+r the digits of B (trgsw.go:51-54); part 0 = A, 1 = B; each [N] is one
 * poly.FourierPoly in the reference's own layout and slot order (poly.go:54-62,
 * fourier_transform.go:64-85,178-247).  The engine permutes it into its wave-native
 * layout on the GPU. */
int tfhe_load_bsk_fourier(tfhe_ctx *ctx, const double *bsk);
/* The same key in the coefficient domain, [n][2L][2][N] uint32 (trgsw.TRGSWLv1,
 * trgsw.go:14-30); the engine runs its own forward FFT (trgsw.go:71-82). */
int tfhe_load_bsk_torus(tfhe_ctx *ctx, const uint32_t *bsk);
/* CloudKey.KeySwitchingKey (cloudkey.go:88-120): [N*t*base][n+1] uint32, flat index
 * base*t*i + base*j + k (keyswitch.go:29).  On the device: the packed rows (the all-zero
 * k = 0 rows dropped) and, for the base-4 sets, a second copy split into byte columns for
 * the matrix-core key switch (78 MB + 104 MB at the 128-bit set). */
int tfhe_load_ksk(tfhe_ctx *ctx, const uint32_t *ksk);

/* cloudkey.NewCloudKey(secretKey) (cloudkey.go:24-31) on the GPU: genBootstrappingKey
 * (cloudkey.go:123-145, trgsw.go:32-82) and genKeySwitchingKey (cloudkey.go:88-120) written
 * straight into the engine's device layouts -- no host-side key, no upload.
 *   s0 [n], s1 [N]: the binary secret keys (key.SecretKey.KeyLv0 / KeyLv1, key.go:10-13)
 *   alpha_lv0 = params.KSKAlpha(), alpha_lv1 = params.BSKAlpha()   (params.go:629-636)
 *   seed128: two 64-bit words, or NULL = 128 bits from the OS entropy source (getrandom), which is what the
 *         reference's auto-seeded math/rand corresponds to.  The same (seed, secret key) pair always gives the
 *         same cloud key (counter-based Philox streams) -- so the seed is SECRET key material: every mask and
 *         noise sample of the published cloud key is a function of it.  Pass fixed seeds in tests only. */
int tfhe_keygen_cloud_seeded(tfhe_ctx *ctx, const uint32_t *s0, const uint32_t *s1, double alpha_lv0,
                             double alpha_lv1, const uint64_t *seed128);
/* The same with a 64-bit test seed (= seed128 {seed, 0}). */
int tfhe_keygen_cloud(tfhe_ctx *ctx, const uint32_t *s0, const uint32_t *s1, double alpha_lv0,
                      double alpha_lv1, uint64_t seed);

/* The loaded keys as opaque device-layout blobs: CloudKey (cloudkey.go:16-21) has no serialised form in the
 * reference; this is the engine's.  which: 0 = bootstrapping key (wave-native spectra, same byte count as the
 * reference's [n][2L][2][N] float64), 1 = key-switching key (packed, zero rows dropped).  A blob is a 64-byte header
 * (magic, device-layout version of the library, which key, the parameter set, payload length, header checksum)
 * followed by the payload; tfhe_key_size = header + payload.  Export copies the blob to caller-owned DEVICE memory of
 * tfhe_key_size bytes (enqueued on `stream`); import takes the blob and ITS LENGTH, checks the header against the
 * context -- another parameter set, the other key, a blob of a library with a different device layout, or a short /
 * long buffer return TFHE_E_INVALID and install nothing -- then installs it (and derives what the engine derives at
 * load time) on `stream`; import synchronises `stream` once, to read the header.  Uses: replicate a key from GPU 0 to
 * the other GPUs of a node with one RCCL broadcast per key instead of N uploads / regenerations, or
 * park it in host memory / on disk between runs. */
int tfhe_key_size(tfhe_ctx *ctx, int which, size_t *bytes);
int tfhe_key_export_dev(tfhe_ctx *ctx, int which, void *d_dst, void *stream);
int tfhe_key_import_dev(tfhe_ctx *ctx, int which, const void *d_src, size_t bytes, void *stream);
/* The same blobs through HOST memory: persist a GPU-generated cloud key, or hand it to another process; synchronous. */
int tfhe_key_export(tfhe_ctx *ctx, int which, void *dst);
int tfhe_key_import(tfhe_ctx *ctx, int which, const void *src, size_t bytes);

/* A replica of `src` -- parameters, dispatch limits and whichever keys it holds -- on GPU device_id: the in-process form of
 * "replicate the read-only keys, shard the batch" (trgsw.BatchBlindRotate fans a batch out over goroutines that SHARE the
 * keys, trgsw.go:234-252; with one key copy per GPU the fan-out needs a replica per device first).  The keys travel GPU to
 * GPU: hipMemcpyPeerAsync between the device layouts (xGMI when hipDeviceCanAccessPeer says the two are peers -- peer access
 * is enabled on the target for the source), a plain device-to-device copy when device_id is the source's own GPU (two
 * contexts on one GPU are two independent submitters); the target then derives what a key load derives.  No host copy of the
 * keys is made.  Fallback when the devices are NOT peers: the copy is staged through 32 MB of page-locked host memory
 * (what tfhe_key_export + tfhe_key_import do, without the full-size host blob).  TFHE_OPT_CLONE_PATH on the new context
 * says which of the three happened.  Synchronous, and ordered behind everything the source has pending: its private stream and
 * every "_dev" call so far, tfhe_key_import_dev included (a clone issued right behind an asynchronous import replicates the
 * complete key).  The source may be used by other threads meanwhile (its keys are immutable
 * after load; key loads on it wait).  The replica is an ordinary context: tfhe_ctx_destroy it.  One goroutine / thread per
 * replica, contiguous shards, results in index order is all the multi-GPU form of gates.Batch* needs (SURVEY.md 8e;
 * shim/go/gpu: CloudKeySet, go-tfhe_amd/host/tfhe_gpu.hpp: cloudkey::CloudKeySet). */
int tfhe_ctx_clone_to(tfhe_ctx *src, int device_id, tfhe_ctx **out);

/* Evaluator.BootstrapAssign / BootstrapLUTAssign over a batch (evaluator.go:139-148,
 * programmable_bootstrap.go:93-115; batch fan-out trgsw.go:234-252).
 *   in      [B][n+1]
 *   testvec [2][N] shared (testvec_per_item = 0) or [B][2][N] (= 1);
 *           NULL = the gate test vector (cloudkey.go:74-85)
 *   out     [B][n+1]
 * Concurrent callers of the host-pointer variant on ONE context are combined exactly as those of tfhe_gate_batch are (see
 * there): the next launch carries every queued request, each item with its caller's table, and each caller gets its rows
 * back, word for word what a call on its own returns -- a combined launch never leaves the kernel shape a lone small call
 * runs (at most one bootstrap per CU at the parameter shapes whose transforms are not exact; calls longer than that, or than
 * TFHE_OPT_COMBINE_MAX, take the context for themselves).  One programmable bootstrap alone: 3.8 ms at the Uint5 set;
 * 64 threads each issuing one: the same 3.8 ms, not 64 x. */
int tfhe_bootstrap_batch(tfhe_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                         int testvec_per_item, uint32_t *out, int B);
int tfhe_bootstrap_batch_dev(tfhe_ctx *ctx, const uint32_t *d_in, const uint32_t *d_testvec,
                             int testvec_per_item, uint32_t *d_out, int B, void *stream);

/* Programmable bootstrap through an EXTENDED lookup table, LookUpTableSize = ext * N (polyExtendFactor = ext):
 * what the Uint6 / Uint7 / Uint8 parameter sets are specified for (params.go:399-402,440-443,481-484) and the
 * reference does not implement (params/UINT_STATUS.md:12-30, uint_params_test.go:29-31 skips them).  The table
 * is a polynomial of degree ext*N over Y^(ext*N) = -1, handed over de-interleaved:
 *   lut [ext][2][N]   component k holds the coefficients of Y^(i*ext + k), i < N (A then B); shared, or
 *                     [B][ext][2][N] with lut_per_item = 1
 * ext = 1 is tfhe_bootstrap_batch.  Every ciphertext word is mod-switched to [0, 2*ext*N) and each CMUX step
 * runs ext external products (one launch per step: a functional path for the experimental sets).  N = 2048
 * parameter shape only; anything else returns TFHE_E_INVALID. */
int tfhe_bootstrap_extended_batch(tfhe_ctx *ctx, const uint32_t *in, const uint32_t *lut, int lut_per_item,
                                  int ext, uint32_t *out, int B);
int tfhe_bootstrap_extended_batch_dev(tfhe_ctx *ctx, const uint32_t *d_in, const uint32_t *d_lut, int lut_per_item,
                                      int ext, uint32_t *d_out, int B, void *stream);

/* Evaluator.BlindRotateAssign / trgsw.BatchBlindRotate (evaluator.go:110-135,
 * trgsw.go:234-252): out_trlwe [B][2][N].  nsteps < 0 means all n CMUX steps; a
 * smaller value stops the chain early (test seam for CMuxAssign, evaluator.go:85-106). */
int tfhe_blind_rotate_batch(tfhe_ctx *ctx, const uint32_t *in, const uint32_t *testvec,
                            int testvec_per_item, uint32_t *out_trlwe, int B, int nsteps);
int tfhe_blind_rotate_batch_dev(tfhe_ctx *ctx, const uint32_t *d_in, const uint32_t *d_testvec,
                                int testvec_per_item, uint32_t *d_out_trlwe, int B, int nsteps,
                                void *stream);

/* Evaluator.ExternalProductAssign (evaluator.go:50-81) of in[b] with bootstrapping-key
 * element bsk[key_index]: in/out [B][2][N]. */
int tfhe_external_product_batch(tfhe_ctx *ctx, int key_index, const uint32_t *in_trlwe,
                                uint32_t *out_trlwe, int B);

/* The decomposition offset the context derived from its parameters (cloudkey.go:60-71, CloudKey.DecompositionOffset): what the
 * reference's callers pass as `decompositionOffset` to trgsw.BlindRotate / BatchBlindRotate (trgsw.go:197,234); a shim compares the
 * caller's value with this one before it routes such a call to tfhe_blind_rotate_batch (which always uses the context's). */
int tfhe_ctx_decomposition_offset(tfhe_ctx *ctx, uint32_t *offset);

/* trgsw.ExternalProductWithFFT (trgsw.go:108-137) / Evaluator.ExternalProductAssign (evaluator.go:50-81) with ANY TRGSW operand --
 * not only an element of the loaded bootstrapping key (tfhe_external_product_batch): the same operand for the whole batch.
 *   trgsw_fourier [2L][2][N] float64: TRLWEFFT[r].A.Coeffs then .B.Coeffs, r < 2L, each in the reference's own FourierPoly layout
 *                 and slot order (trgsw.go:60-68, poly.go:54-62) -- one element of what tfhe_load_bsk_fourier takes n of
 *   decomposition_offset: the caller's `decompositionOffset` argument (any value: it is a kernel operand, not context state)
 *   in / out      [B][2][N]
 * Needs no loaded key.  Exactness follows the parameter shape as everywhere. */
int tfhe_external_product_with(tfhe_ctx *ctx, const double *trgsw_fourier, uint32_t decomposition_offset,
                               const uint32_t *in_trlwe, uint32_t *out_trlwe, int B);
/* trgsw.CMUX(in1, in2, cond) (trgsw.go:173-194) / Evaluator.CMuxAssign(ctCond, ct0, ct1) (evaluator.go:85-106):
 * out[b] = ct0[b] + cond (x) (ct1[b] - ct0[b]), i.e. ct0 where cond encrypts 0 and ct1 where it encrypts 1.  Operand as above. */
int tfhe_cmux_with(tfhe_ctx *ctx, const double *trgsw_fourier, uint32_t decomposition_offset, const uint32_t *ct0_trlwe,
                   const uint32_t *ct1_trlwe, uint32_t *out_trlwe, int B);

/* trlwe.SampleExtractIndex / SampleExtractIndexAssign for ANY index k in [0, N) (trlwe.go:114-128, trlwe_ops.go:10-21):
 * in [B][2][N] -> out [B][N+1] (a tlwe.TLWELv1: N mask words, body last).  The bootstrap itself only ever uses k = 0, fused into the
 * key switch (tfhe_extract_keyswitch_batch); this is the seam on its own. */
int tfhe_sample_extract_batch(tfhe_ctx *ctx, const uint32_t *in_trlwe, int k, uint32_t *out_lwe1, int B);
/* trgsw.IdentityKeySwitching / IdentityKeySwitchingAssign (trgsw.go:285-312, keyswitch.go:10-37) on already-extracted samples:
 * in [B][N+1] (tlwe.TLWELv1) -> out [B][n+1].  Runs the same key-switch kernels as the fused form (bit-exact with it). */
int tfhe_keyswitch_batch(tfhe_ctx *ctx, const uint32_t *in_lwe1, uint32_t *out, int B);

/* trlwe.SampleExtractIndexAssign(.,0,.) + trgsw.IdentityKeySwitchingAssign
 * (trlwe_ops.go:10-21, keyswitch.go:10-37): in [B][2][N] -> out [B][n+1]. */
int tfhe_extract_keyswitch_batch(tfhe_ctx *ctx, const uint32_t *in_trlwe, uint32_t *out, int B);
int tfhe_extract_keyswitch_batch_dev(tfhe_ctx *ctx, const uint32_t *d_in_trlwe, uint32_t *d_out,
                                     int B, void *stream);

/* gates.NAND ... gates.ORYN, gates.MUX and gates.Batch* (gates.go:26-114,156-312):
 * the linear preparation (gates_helper.go:10-63) is fused into the bootstrap kernel.
 *   ops: per-item op codes [B], or NULL with op_uniform = one TFHE_OP_* for all items
 *   a, b: [B][n+1]; c: [B][n+1], required iff any op is TFHE_OP_MUX (3 bootstraps,
 *   gates.go:107-114), may be NULL otherwise.
 * Batch XNOR follows the tested scalar gates.XNOR (+1/4, gates.go:52-58), not
 * BatchXNOR's -1/4 (gates.go:293), which computes XOR.
 * MUX items are found and compacted on the device (no op code is ever copied back): one blind-rotate request
 * covers every item's own gate (a MUX item's AND(a,b)) plus ANDNY(a,c) of the MUX items, a second one their OR.
 * The host-pointer variant validates the op codes before issuing anything; the _dev variant cannot (see
 * tfhe_ctx_sync).
 *
 * Concurrent callers of the host-pointer variant (any number of threads on ONE context) are COMBINED, not serialised: a
 * launch of 1 ... one bootstrap per CU costs the same ~2.2 ms, so while one launch is in flight the other callers claim rows of the
 * next batch's page-locked staging (lock-free; each copies its own operands in) and the next launch carries ALL of them as one gate
 * batch with per-item op codes; each caller takes exactly its rows out -- bit-identical to what a call on its own returns (a gate's
 * result depends on its own operands only; at the parameter shapes whose transforms are not exact a combined launch stays within
 * the kernel shape a lone small call runs, a MUX row counting as two bootstraps).  There is no extra thread and a lone caller is
 * launched at once; the caller that leads a batch right behind a combined launch waits for that launch's callers to come back
 * (ends when all are back, after TFHE_OPT_COMBINE_QUIET_US without a new arrival, or after four such windows).  A combined launch
 * carries at most one row per CU; calls of more than TFHE_OPT_COMBINE_MAX gates (default: the same) take the context for themselves.
 * (The reference's scalar gates.* share one evaluator that is not goroutine-safe, gates.go:19-23; its concurrency is one pooled
 * evaluator per goroutine, trgsw.go:227-252 -- this is what replaces it.)  If a combined launch fails for a reason only the
 * combination has, its calls are re-issued one by one, each with its own result.
 * Larger host-pointer calls (more than one gate per CU, up to 16 launches' worth) of several threads are not combined but OVERLAPPED:
 * each works in one of two buffer slots, and one caller's upload / download runs while another's kernels do (the kernels themselves run
 * one call at a time: same words as a lone call).  Longer calls pipeline their own transfers slab by slab. */
int tfhe_gate_batch(tfhe_ctx *ctx, const uint8_t *ops, int op_uniform, const uint32_t *a,
                    const uint32_t *b, const uint32_t *c, uint32_t *out, int B);
int tfhe_gate_batch_dev(tfhe_ctx *ctx, const uint8_t *d_ops, int op_uniform, const uint32_t *d_a,
                        const uint32_t *d_b, const uint32_t *d_c, uint32_t *d_out, int B,
                        void *stream);

/* poly.Evaluator.ToFourierPolyAssign / ToPolyAssignUnsafe over a batch of polynomials
 * (fourier_transform.go:18-21,40-44), spectra in the reference FourierPoly layout:
 * polys [P][N] uint32 <-> spectra [P][N] float64.  Test seams for the FFT. */
int tfhe_to_fourier_batch(tfhe_ctx *ctx, const uint32_t *polys, double *spectra, int P);
int tfhe_to_poly_batch(tfhe_ctx *ctx, const double *spectra, uint32_t *polys, int P);

/* Elapsed GPU time (ms) of the most recent blind-rotate kernel launch on this context,
 * measured with HIP events on the stream it ran on; blocks until it has finished.
 * which: 0 = blind rotate, 1 = sample-extract + key switch. */
int tfhe_last_kernel_ms(tfhe_ctx *ctx, int which, float *ms);

/* Page-locked host buffers for the host-pointer entry points.  The Go shim flattens ciphertexts
 * anyway (cgo cannot pass []*TLWELv0); flattening INTO a buffer from tfhe_host_alloc lets the
 * transfers run as true asynchronous DMA at PCIe speed instead of through the runtime's pageable
 * staging path (measured, 1024 NAND gates: 7.5 ms pageable, 7.1 ms page-locked, 6.9 ms with
 * device-resident operands).  Plain malloc'ed pointers remain valid everywhere. */
int tfhe_host_alloc(size_t bytes, void **out);
int tfhe_host_free(void *p);

/* Cumulative per-kernel timing for benchmarks: while enabled, every launch of the two path
 * kernels is bracketed by its own HIP event pair on the stream it is launched on.
 * tfhe_timing_read blocks until those launches have finished, returns their count and summed
 * duration, and clears the list.  which: 0 = blind rotate, 1 = extract + key switch. */
int tfhe_timing_enable(tfhe_ctx *ctx, int on);
int tfhe_timing_read(tfhe_ctx *ctx, int which, int *launches, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif
