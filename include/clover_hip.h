/*
 * clover_hip.h -- C ABI of libclover_hip.so: the MI355X (gfx950) backend for Clover's 4-bit hot path.
 *
 * This is the drop-in boundary.  Clover itself has no FFI/plugin interface: its API is the public method
 * surface of the header-only classes CloverVector4 / CloverMatrix4 (reference README.md:68, 117-128).
 * Each entry point below replaces the BODY of one of those methods; the reference-side binding a
 * maintainer would add is shown in INTEGRATION.md, and include/CloverVector4.h / include/CloverMatrix4.h
 * in this repository are C++ containers with the reference's class and method names built on this ABI.
 *
 * Conventions
 *  - plain C, no C++/torch types.  Every data pointer is a DEVICE pointer (HBM) unless the name says
 *    `host`; `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - sizes are the PADDED sizes the reference containers hold: vector length_pad (multiple of 128,
 *    CloverVector.h:86-92), matrix rows/cols (multiples of 128, CloverMatrix.h:48-53).  The mvm family (clm4_mvm,
 *    clm4_rowdots, clm4_mvm_scale_and_add, clm4_mvm_v8*, clm4_rowdots_v8, clm4_mvm_f32) also accepts rows % 64 == 0:
 *    a row shard of a matrix, the unit mvm_parallel gives a thread (CloverMatrix4.h:1700-1705).
 *  - data format = the reference's: byte i holds element 2i in its HIGH nibble and 2i+1 in its LOW nibble,
 *    two's complement, values in [-7,7] (CloverVector4.h:511-514); one fp32 scale (the block's absolute
 *    maximum, 0 -> 1.0) per 64 elements (CloverVector4.h:661-673) / per 64x64 tile, row-major tile grid
 *    (CloverMatrix4.h:123-139, 598-603).  Matrices are row-major nibbles, exactly as the reference
 *    stores them.
 *  - every function returns CLV_OK (0) or a negative CLV_ERR_* code and never calls exit(); the C++
 *    container layer turns a non-zero status into the reference's behaviour (message + exit(1)).
 *    clv_last_error() returns a thread-local description of the last failure.
 *  - results: bit-identical to the reference's AVX2 path when stochastic rounding is disabled
 *    (rng == NULL); with an rng state the same XORShift stream as the reference's sequential methods is
 *    consumed (bit-identical nibbles for identical keys).  Exceptions are spelled out per function.
 *  - STOCHASTIC CALLS (any function given a non-NULL rng_state_dev: clv4_quantize, clm4_quantize, clm4_mvm, clv4_scale_and_add,
 *    clm4_mvm_scale_and_add, clv8_quantize, clv8_scale_and_add, clm4_mvm_v8, clm4_mvm_v8_scale_and_add, clm4_iht, clm4_iht_v8):
 *      (1) calls that share a state buffer must be STREAM-ORDERED -- the same stream, or an event / sync between them: they consume one
 *          sequential XORShift stream, as the reference's methods do on one object, and each launch reads the state its predecessor left;
 *      (2) such a call must NOT be captured into a hipGraph unless its state is in graph mode (clv_rng_graph_mode): ordinarily every launch
 *          carries a fresh host-side sequence number as a kernel argument (how a kernel tells the state slot written by its predecessor
 *          from the one it writes itself), and a replayed graph would replay the number.  Deterministic calls (rng_state_dev == NULL)
 *          capture fine;
 *      (3) a state belongs to the process that initialised it (re-key with clv_rng_set after sharing the buffer across processes).
 *  - asynchrony: every call only enqueues work on `stream`; results are valid after clv_stream_sync / an event.
 *    clv4_dot, the threshold functions (when `workspace` is NULL) and clm4_gemm use grow-only scratch owned by the library, one
 *    buffer per (device, stream): calls on different streams never share scratch and may overlap.  clv4_dot / clv8_dot in CLV_DOT_FAST
 *    mode are ONE launch whose workgroups hand their partials over through 64 KiB of zero-initialised slots, also per (device, stream),
 *    allocated (hipMalloc + a memset on the stream) by the first such call there: make one ordinary call on a stream before capturing
 *    it into a hipGraph (tests/test_graph_capture.py; a first call under capture returns CLV_ERR_INVALID with that advice); the kernel
 *    leaves the slots zero, so replays need nothing else.  Two limits of the single-launch form: (a) a captured graph holds the slot
 *    pointer of the stream it was captured on -- do not replay it on another stream WHILE that stream runs a FAST dot (two kernels
 *    would share slots); (b) the collecting workgroup (the grid's last) assumes every other workgroup of the grid has been dispatched
 *    when it runs: true of this hardware's dispatchers, not a guarantee of the HIP model -- its wait is bounded (4 s, then a trap) so
 *    that an anomaly is an error, not a hang.
 *  - clm4_iht runs Q_IHT / Q_GD as ONE persistent launch when the problem qualifies (rounding disabled, threshold FAST or none, m and
 *    n <= 8192 and both matrices fit the chip's LDS): every workgroup of its grid must be resident at once, so (a) two such launches on
 *    different streams of one device are chained by an event (the second waits for the first; no host blocking), (b) ANOTHER PROCESS
 *    running such a launch on the same device at the same time is not covered -- the kernel's waits are bounded (4 s, then a trap),
 *    (c) CLV_IHT_PERSISTENT=0 in the environment (read per call) selects the launch-per-step loop.  Same bits either way.
 */
#ifndef CLOVER_HIP_H
#define CLOVER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLV_OK                 0
#define CLV_ERR_INVALID       -1   /* bad size / null pointer / unsupported mode */
#define CLV_ERR_HIP           -2   /* a HIP runtime call failed (see clv_last_error) */
#define CLV_ERR_NO_DEVICE     -3   /* no gfx950 device visible */
#define CLV_ERR_UNSUPPORTED   -4

/* dot() evaluation order */
#define CLV_DOT_EXACT          0   /* the reference's 16 sequential fma chains: bit-identical, latency-bound */
#define CLV_DOT_FAST           1   /* exact integer block sums, fp32 tree reduction: bandwidth-bound */

/* ---- runtime ---------------------------------------------------------------------------------- */
const char *clv_version(void);
const char *clv_last_error(void);
int  clv_device_count(int *count);
int  clv_set_device(int device);
int  clv_get_device(int *device);
/* name, compute units, HBM bytes of the current device (any pointer may be NULL) */
int  clv_device_info(char *name, int name_len, int *compute_units, uint64_t *hbm_bytes);

int  clv_malloc(void **ptr, uint64_t bytes);
int  clv_free(void *ptr);
int  clv_memset(void *ptr, int value, uint64_t bytes, void *stream);
int  clv_memcpy_h2d(void *dst_dev, const void *src_host, uint64_t bytes, void *stream);
int  clv_memcpy_d2h(void *dst_host, const void *src_dev, uint64_t bytes, void *stream);
int  clv_memcpy_d2d(void *dst_dev, const void *src_dev, uint64_t bytes, void *stream);
int  clv_host_alloc(void **ptr, uint64_t bytes);          /* pinned, page-aligned host memory */
int  clv_host_free(void *ptr);
int  clv_stream_create(void **stream);
int  clv_stream_destroy(void *stream);
int  clv_stream_sync(void *stream);
int  clv_device_sync(void);
int  clv_event_create(void **event);
int  clv_event_destroy(void *event);
int  clv_event_record(void *event, void *stream);
int  clv_event_sync(void *event);
int  clv_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- XORShift state (CloverRandom.h:90-114, simdxorshift128plus.h:47-109) ------------------------ */
/* The generator state lives in a CLV_RNG_STATE_BYTES device buffer: 8 x uint64 = s0[4], s1[4] (random_key1,
 * random_key2), kept twice plus launch stamps so that a kernel can advance the state inside its own launch.
 * Initialise it with clv_rng_seed() (reproduces avx_xorshift128plus_init(key1, key2) on the host and uploads
 * it) or clv_rng_set() (explicit keys, CloverRandom::setRandomKeys); never write the buffer directly.
 * Calls that share a state must be ordered (same stream, or synchronised) -- they consume one sequential
 * stream -- and must not be captured into a hipGraph unless the state is in graph mode (clv_rng_graph_mode below: each call otherwise
 * carries a fresh host-side launch stamp).  A state belongs to the
 * process that initialised it (the stamps come from a per-process counter): re-key with clv_rng_set after sharing a buffer. */
#define CLV_RNG_STATE_BYTES 256
int  clv_rng_seed(uint64_t *state_dev, uint64_t key1, uint64_t key2, void *stream);
int  clv_rng_set(uint64_t *state_dev, const uint64_t key1[4], const uint64_t key2[4], void *stream);
int  clv_rng_get(const uint64_t *state_dev, uint64_t key1[4], uint64_t key2[4], void *stream); /* syncs */
/* Graph mode (opt-in, per state): on != 0 moves the launch stamps of this state from the host's counter into the state buffer itself --
 * every stochastic call on it then enqueues a one-thread "tick" kernel in front of its own kernel (about 2 us of stream time) and carries
 * no per-launch argument any more, so the calls CAN be captured into a hipGraph and the graph replayed: each replay consumes the next
 * part of the XORShift stream, exactly as the same calls issued one after another would.  Enable it before capturing (outside the
 * capture), on the stream the calls will use; on == 0 returns the state to host stamps.  clv_rng_set / clv_rng_seed keep the mode. */
int  clv_rng_graph_mode(uint64_t *state_dev, int on, void *stream);
/* Tuning knob (process-wide): how many 8-block segments of the stream one wavefront of the stochastic VECTOR kernels walks -- 1, 4, 16,
 * 64 (= 16 segments of 32 blocks), or 0 = by size (the default).  The choice never changes a result, only the speed at a given length;
 * the parity tests use it to take every kernel shape through the small cases.  Environment: CLV_ST_SEGMENTS. */
int  clv_rng_set_segments(int segments);

/* ---- CloverVector4 ---------------------------------------------------------------------------- */
/* CloverVector4::quantize (CloverVector4.h:605-807).  x: n_pad floats; q: n_pad/2 bytes; s: n_pad/64
 * floats.  rng_state_dev == NULL <=> CLOVER_STOCHASTIC_ROUNDING_DISABLED; otherwise two draws per block
 * are consumed in block order and the state is advanced in place, as the sequential method does (with an rng: order the calls
 * that share the state on one stream, and do not capture them in a hipGraph -- "STOCHASTIC CALLS" above). */
int  clv4_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng_state_dev, void *stream);
/* CloverVector4::restore (CloverVector4.h:1027-1093). */
int  clv4_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x, void *stream);
/* CloverVector4::dot (CloverVector4.h:1095-1192).  *out_dev receives one float.  CLV_DOT_EXACT is
 * bit-identical to the reference; CLV_DOT_FAST differs only in fp32 summation order (the per-block
 * integer sums are exact either way; one launch, deterministic: a fixed tree whatever order the workgroups finish in).
 * workspace: clv4_dot_workspace_bytes() bytes or NULL (internal); only CLV_DOT_EXACT uses it. */
uint64_t clv4_dot_workspace_bytes(uint64_t n_pad);
int  clv4_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad,
              int mode, float *out_dev, void *workspace, void *stream);
/* exact int32 sum of the 8 nibble products of every 32-bit word: I[n_pad/8] */
int  clv4_word_isums(const int8_t *qu, const int8_t *qv, uint64_t n_pad, int32_t *isums, void *stream);

/* ---- CloverMatrix4 ---------------------------------------------------------------------------- */
/* CloverMatrix4::quantize (CloverMatrix4.h:512-766).  A: rows*cols floats row-major; q: rows*cols/2
 * bytes; s: (rows/64)*(cols/64) floats.  With an rng the tiles consume the stream in the reference's
 * order (column-block outer, row-block inner, two draws per tile row).  With an rng: stream-ordered per state, no graph capture. */
int  clm4_quantize(const float *A, uint64_t rows, uint64_t cols, int8_t *q, float *s,
                   uint64_t *rng_state_dev, void *stream);
/* CloverMatrix4::restore_scalar (CloverMatrix4.h:266-301): A[i][j] = f32(s_tile/7) * q, rows*cols floats row-major */
int  clm4_restore(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, float *A, void *stream);
/* CloverMatrix4::mvm(const CloverVector4&, CloverVector4&) (CloverMatrix4.h:777-1083) and mvm_parallel
 * (:1681-2006; same results).  x: cols/2 bytes + cols/64 scales; r: rows/2 bytes + rows/64 scales
 * (the re-quantised result).  Bit-identical to the reference when rng_state_dev == NULL; with an rng the re-quantisation draws two
 * values per output block from the state (lane map 8j+g, CloverMatrix4.h:925-932): stream-ordered per state, no graph capture. */
int  clm4_mvm(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols,
              const int8_t *x, const float *sx, int8_t *r, float *sr,
              uint64_t *rng_state_dev, void *stream);
/* the fp32 row dots of mvm before re-quantisation: d[rows] (CloverMatrix4.h:804-916) */
int  clm4_rowdots(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                  const int8_t *x, const float *sx, float *d, void *stream);
/* GEMM, build-defined (the reference has none; semantics in oracle/clover4_oracle.h and DESIGN.md):
 * A is M x K, B is N x K, C = A * B^T as fp32 M x N row-major, one fma chain over K-blocks per element.
 * Uses grow-only scratch of (M + N) * K * 3/4 bytes that belongs to (device, stream): calls on different streams may overlap. */
int  clm4_gemm(const int8_t *A, const float *sA, uint64_t M, uint64_t K,
               const int8_t *B, const float *sB, uint64_t N, float *C, void *stream);

/* The matrix kernel behind clm4_gemm streams its operands as FP6 (E2M3) codes in staging order; clm4_gemm re-codes both of them on
 * every call.  An operand that is multiplied many times (weights) can be re-coded once: clm4_gemm_prepare allocates rows*K*3/4
 * bytes of HBM for the image (it keeps no reference to q: prepare again after q changes), clm4_gemm_prepared takes either operand
 * prepared (op != NULL) or raw (op == NULL, nibbles in A / B).  Results are those of clm4_gemm, bit for bit.  An image's staging layout
 * (128- or 256-row tiles) is chosen from the operand's own row count; when two prepared operands disagree (one has >= 4096 rows on a
 * 256-CU part, the other fewer) the smaller one is re-coded inside the call, for which its nibbles must be passed besides its image
 * (A / B may always be given next to opA / opB; CLV_ERR_INVALID if they are needed and missing). */
typedef struct clm4_gemm_operand clm4_gemm_operand;
int  clm4_gemm_prepare(const int8_t *q, uint64_t rows, uint64_t K, clm4_gemm_operand **op, void *stream);
int  clm4_gemm_release(clm4_gemm_operand *op);
int  clm4_gemm_prepared(const clm4_gemm_operand *opA, const int8_t *A, const float *sA, uint64_t M, uint64_t K,
                        const clm4_gemm_operand *opB, const int8_t *B, const float *sB, uint64_t N, float *C, void *stream);
/* The exact integer part of the GEMM (output (1)): S[i][j] = sum over K-blocks [kb_begin, kb_begin + kb_count) of the
 * 64 nibble products, int32, row-major M x N.  With the full range this is the unscaled int4 x int4 -> int32 GEMM; with a range of
 * one K-block it is the per-block sum the fp32 result folds.  Even kb_begin and kb_count run on the matrix kernel (accumulating
 * across K-blocks inside the pipe: integers below 2^24 are exact there), other ranges on a VALU kernel. */
int  clm4_gemm_i32(const int8_t *A, uint64_t M, uint64_t K, const int8_t *B, uint64_t N, uint64_t kb_begin, uint64_t kb_count,
                   int32_t *S, void *stream);
/* the same with either operand prepared once (clm4_gemm_prepare; op == NULL: raw nibbles in A / B).  An odd K-block range needs the
 * nibbles of both operands besides any prepared image. */
int  clm4_gemm_i32_prepared(const clm4_gemm_operand *opA, const int8_t *A, uint64_t M, uint64_t K,
                            const clm4_gemm_operand *opB, const int8_t *B, uint64_t N, uint64_t kb_begin, uint64_t kb_count,
                            int32_t *S, void *stream);

/* ---- callers either side of the hot path: the other steps of the quantized IHT/GD loops ---- */
/* CloverVector4::scaleAndAdd (CloverVector4.h:1196-1478; _parallel :1489-1791): r = quantize(u + a*v) per
 * 64-block; r/sr may alias qu/su (the in-place overload).  Bit-identical; with an rng the sequential
 * method's XORShift stream and its lane map (element 8j+(g^1)) are reproduced (stream-ordered per state, no graph capture). */
int  clv4_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad,
                        int8_t *r, float *sr, uint64_t *rng_state_dev, void *stream);
/* mvm immediately followed by scaleAndAdd on its result -- the pairs "t2 = y - Phi x" and "x += mu Phi' t2" of the
 * quantized IHT / GD loops (test/performance/01_measure.h:923-946, 999-1021) -- in one launch:
 *     t = quantize(A x)  (stored if t/st != NULL),   r = quantize(u + a * t);   u, r: rows elements.
 * Bit-identical to clm4_mvm(...) followed by clv4_scale_and_add(u, t, a, r), including the XORShift stream positions
 * when an rng is given.  r/sr may alias u/su; they must not alias x/sx. */
int  clm4_mvm_scale_and_add(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                            const int8_t *qu, const float *su, float a, int8_t *t, float *st, int8_t *r, float *sr,
                            uint64_t *rng_state_dev, void *stream);
/* ---- mixed precision: 4-bit matrix x 8-bit vector --------------------------------------------- */
/* CloverVector8 (CloverVector8.h:35-140): n_pad int8 values in natural order + n_pad/64 fp32 scales, value = q*scale/127.
 * clv8_quantize = CloverVector8::quantize (:393-606), clv8_restore = ::restore (:835-909); bit-identical, and with an
 * rng the same XORShift stream positions and lane map as the reference (two draws per 64-block). */
int  clv8_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng_state_dev, void *stream);
int  clv8_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x, void *stream);
/* CloverVector8::scaleAndAdd (CloverVector8.h:1063-1358) and ::threshold (:1680-1740): the vector steps of the quantized
 * IHT / GD loops in the configuration the reference publishes for "4-bit" (CloverMatrix4 with CloverVector8 vectors,
 * test/performance/02_bit04.cpp:140).  Same contracts as clv4_scale_and_add / clv4_threshold. */
int  clv8_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad,
                        int8_t *r, float *sr, uint64_t *rng_state_dev, void *stream);
/* clv8_dot = CloverVector8::dot (CloverVector8.h:911-977): per block the 64 byte products summed exactly per 32-bit lane of the block's two
 * halves, scale = f32(f32(su * 1/127) * f32(sv * 1/127)), 8 fma chains over ALL blocks, then the _mm256_haddf32_ps tree.  CLV_DOT_EXACT: that
 * order, bit for bit (n/64 dependent fmas per chain: latency-bound by definition); CLV_DOT_FAST (= dot_parallel, :979-1061): the same exact
 * block integers, fp32 tree order, one launch, memory-bound.  workspace: clv8_dot_workspace_bytes() bytes or NULL (internal; EXACT only). */
uint64_t clv8_dot_workspace_bytes(uint64_t n_pad);
int  clv8_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad, int mode, float *out_dev,
              void *workspace, void *stream);
uint64_t clv8_threshold_workspace_bytes(uint64_t n_pad);
int  clv8_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream);
int  clv8_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream);
/* CloverMatrix4::mvm(const CloverVector8 &, CloverVector8 &) (CloverMatrix4.h:1093-1441; _parallel :2017-2387):
 * x: cols int8 + cols/64 scales; r: rows int8 + rows/64 scales (re-quantised to 8 bits).  Bit-identical to the
 * reference's SIMD path (8 fp32 fma chains per row) for either rounding mode. */
int  clm4_mvm_v8(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                 int8_t *r, float *sr, uint64_t *rng_state_dev, void *stream);
/* clm4_mvm_v8 immediately followed by clv8_scale_and_add on its result, one launch (see clm4_mvm_scale_and_add):
 * t = quantize8(A x) (stored if t/st != NULL), r = quantize8(u + a * t); r/sr may alias u/su, not x/sx. */
int  clm4_mvm_v8_scale_and_add(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                               const int8_t *qu, const float *su, float a, int8_t *t, float *st, int8_t *r, float *sr,
                               uint64_t *rng_state_dev, void *stream);
/* Q_IHT / Q_GD with CloverMatrix4 and CloverVector8 vectors -- the reference's published "4-bit" IHT configuration
 * (test/performance/02_bit04.cpp:140, doc/results/performance.txt:597-606).  Arguments as clm4_iht; vectors are 8-bit. */
int  clm4_iht_v8(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n,
                 int8_t *x, float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1,
                 int8_t *t2, float *st2, int8_t *t3, float *st3, uint64_t iterations, uint64_t K, float mu, int threshold,
                 uint64_t *rng_state_dev, void *stream);
/* diagnostic: the number of clm4_iht / clm4_iht_v8 calls of this process that ran as ONE persistent launch (iht_persist.hip) rather than
 * as the launch-per-step loop -- what a test or a benchmark reads to know which of the two it measured. */
uint64_t clv_iht_persistent_launches(void);
/* the fp32 row dots of that mvm before re-quantisation: d[rows] (CloverMatrix4.h:1120-1243) */
int  clm4_rowdots_v8(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                     float *d, void *stream);
/* CloverVector4::threshold(K) (CloverVector4.h:1913-2060): keep the K largest |value| among the first n
 * elements, zero the other nibbles in place.  The surviving multiset of magnitudes equals the reference's;
 * among EQUAL magnitudes the lowest indices survive (the reference's choice depends on its heap order).
 * Vectors beyond one workgroup (n_pad > 131072) take three launches, the middle one a persistent kernel (one resident workgroup per CU)
 * that hands its radix levels over through a zero-initialised control block of the (device, stream), allocated -- like the hand-over
 * slots of the single-launch dots -- by the first such call there: a FIRST call made inside a stream capture runs the older six-launch
 * form instead (same result); after one ordinary call the three launches capture and replay (tests/test_threshold_large3.py).
 * `workspace` (or NULL: library scratch of the stream) needs clv4_threshold_workspace_bytes(n_pad) bytes and no initialisation. */
uint64_t clv4_threshold_workspace_bytes(uint64_t n_pad);
int  clv4_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream);
/* The same with the tie rule chosen by `mode` (the threshold counterpart of clv4_dot's CLV_DOT_EXACT / CLV_DOT_FAST):
 *   CLV_THRESHOLD_FAST       radix select, lowest-index ties (= clv4_threshold; bandwidth-bound);
 *   CLV_THRESHOLD_REFERENCE  the reference's survivor SET, index for index: its K-entry min-heap walk (CloverVector4.h:1927-1972,
 *                            std::make_heap under gt_idx_t + min_heapify, CloverBase.h:208-249) is reproduced step by step by one
 *                            wavefront with the heap in LDS.  Sequential by definition (which of several equal magnitudes survive
 *                            depends on the whole insertion history): about 0.4 us per heap insert -- N = 8192, K = 1024: 0.85 ms.
 *                            The C++ containers take this mode by default (clover_device.h: the exactness switch); here it is a mode.
 * `workspace` (NULL = library scratch of the stream, sized for the k of the call) needs clv_threshold_reference_workspace_bytes_k(n_pad, k)
 * in REFERENCE mode.
 * clm4_iht / clm4_iht_v8 take threshold = 2 for this mode (1 = FAST, 0 = no threshold: Q_GD). */
#define CLV_THRESHOLD_FAST 0
#define CLV_THRESHOLD_REFERENCE 1
uint64_t clv_threshold_reference_workspace_bytes(uint64_t n_pad);                    /* enough for any k */
uint64_t clv_threshold_reference_workspace_bytes_k(uint64_t n_pad, uint64_t k);      /* for this k: the 8 (k + 1)-byte heap region only when k > 20000 */
int  clv4_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream);
/* CloverVector4::threshold_min_heap(idx_t *min_heap, uint64_t k) (CloverVector4.h:1929-1970; CloverVector8.h:1696-1737): the REFERENCE
 * mode above, and the K-entry heap the reference leaves in the caller's memory: heap_dev[i] = {fp32 |value|, uint32 element index},
 * i < k (8 k bytes of device memory), entry for entry in the reference's array order after the walk.  1 <= k <= n. */
int  clv4_threshold_heap(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *heap_dev, void *workspace, void *stream);
int  clv8_threshold_heap(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *heap_dev, void *workspace, void *stream);
/* CloverMatrix4::transpose (CloverMatrix4.h:1549-1663; _parallel :2508-2640): qt(j,i) = q(i,j), tile scales
 * transposed.  q is rows x cols, qt is cols x rows.  Exact. */
int  clm4_transpose(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, int8_t *qt, float *st, void *stream);

/* mixed precision CloverMatrix4::mvm(const CloverVector32&, CloverVector32&) (CloverMatrix4.h:1451-1547):
 * x: cols floats, r: rows floats (fp32 row dots, no re-quantisation).  Bit-identical (32 fma chains per row). */
int  clm4_mvm_f32(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const float *x, float *r, void *stream);
/* Q_IHT / Q_GD (test/performance/01_measure.h:923-946, 999-1021): x.clear(), then `iterations` times
 *   t1 = Phi*x; t2 = y - t1; t3 = PhiT*t2; x = x + mu*t3; [threshold(K)]          (threshold != 0: IHT, else GD)
 * entirely on the device.  Phi is m x n, PhiT its transpose (n x m), x has n (padded) / x_len (logical)
 * elements, y, t1, t2 have m, t3 has n.  All launches are enqueued on `stream`; nothing is copied back. */
int  clm4_iht(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n,
              int8_t *x, float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1,
              int8_t *t2, float *st2, int8_t *t3, float *st3, uint64_t iterations, uint64_t K, float mu, int threshold,
              uint64_t *rng_state_dev, void *stream);

/* ---- multi-GPU: row-sharded mvm on the GPUs of one node (one process, RCCL over xGMI) --------------- */
/* MI355X counterpart of mvm_parallel's contiguous split of 64-row blocks over threads
 * (CloverMatrix4.h:1700-1705): shard `part` owns a contiguous multiple of 64 rows, x is replicated, the
 * packed result is all-gathered; no partial sums cross devices, so results equal clm4_mvm bit for bit. */
typedef struct clm4_shard_ctx clm4_shard_ctx;
int  clm4_shard_partition(uint64_t rows, int nparts, int part, uint64_t *row_begin, uint64_t *row_count);
/* devices: NULL = 0..ndev-1.  Listing a device more than once is allowed (all shard arithmetic on fewer GPUs, exchanges become
 * plain copies, no RCCL): the layout tests use to run ragged and 8-way partitions on one GPU. */
int  clm4_sharded_create(clm4_shard_ctx **ctx, int ndev, const int *devices, uint64_t rows, uint64_t cols);
int  clm4_sharded_destroy(clm4_shard_ctx *ctx);
int  clm4_sharded_info(const clm4_shard_ctx *ctx, int part, int *device, uint64_t *row_begin, uint64_t *row_count,
                       int8_t **A_dev, float **sA_dev);
/* scatter a whole matrix in the reference layout from host memory */
int  clm4_sharded_upload(clm4_shard_ctx *ctx, const int8_t *A_host, const float *sA_host);
/* synthetic matrix: the bytes clv_fill_random_nibbles/scales(seed, seed+1) give for the unsharded matrix */
int  clm4_sharded_fill_random(clm4_shard_ctx *ctx, uint64_t seed);
/* r = A*x; every device ends with the full packed result; r_host/sr_host (optional) receive a copy */
int  clm4_sharded_mvm(clm4_shard_ctx *ctx, const int8_t *x, const float *sx, int x_on_host, int8_t *r_host, float *sr_host);
int  clm4_sharded_result(const clm4_shard_ctx *ctx, int part, const int8_t **r_dev, const float **sr_dev);
/* what the last clm4_sharded_mvm spent on shard `part`: its kernel and the exchange behind it (HIP events on that shard's stream) */
int  clm4_sharded_timing(const clm4_shard_ctx *ctx, int part, float *mvm_ms, float *gather_ms);
/* ranks in the RCCL communicator the context built (0: none needed), and whether the gather is the single ncclAllGather pair
 * (equal shards) or the per-owner broadcasts (ragged shards) */
int  clm4_sharded_comm_info(const clm4_shard_ctx *ctx, int *rccl_ranks, int *equal_shards);
/* The same sharded mvm for loops (no host synchronisation inside; clm4_sharded_mvm above is the blocking convenience form):
 *   clm4_sharded_set_x        replicate x (host memory or device `part 0`) to every device, stream-ordered;
 *   clm4_sharded_loop_begin   once: second result buffer + exchange stream per device, `slots` timed steps' worth of events;
 *   clm4_sharded_mvm_enqueue  step `step`: every shard's kernel into result buffer step & 1 on its compute stream, the all-gather of
 *                             that buffer on its exchange stream -- the gather of step i overlaps the kernel of step i + 1;
 *                             timed != 0 (step < slots) keeps the step's events for clm4_sharded_step_timing;
 *   clm4_sharded_sync         waits for everything enqueued on every device;
 *   clm4_sharded_result_buf   device pointers of the full result in buffer `buf` held by shard `part`. */
int  clm4_sharded_set_x(clm4_shard_ctx *ctx, const int8_t *x, const float *sx, int x_on_host);
int  clm4_sharded_loop_begin(clm4_shard_ctx *ctx, int slots);
int  clm4_sharded_mvm_enqueue(clm4_shard_ctx *ctx, int step, int timed);
int  clm4_sharded_sync(clm4_shard_ctx *ctx);
int  clm4_sharded_step_timing(const clm4_shard_ctx *ctx, int part, int step, float *mvm_ms, float *gather_ms);
int  clm4_sharded_result_buf(const clm4_shard_ctx *ctx, int part, int buf, const int8_t **r_dev, const float **sr_dev);
/* C = A * B^T with the sharded A and an N x cols CloverMatrix4 B (host memory or device `part 0`) replicated on every device:
 * device d ends with its rows of C (fp32, N columns), bit-identical to clm4_gemm on the whole matrix; nothing is exchanged
 * between the shards.  Every shard must be a multiple of 128 rows.  C_host (optional) receives the whole C. */
int  clm4_sharded_gemm(clm4_shard_ctx *ctx, const int8_t *B, const float *sB, uint64_t N, int b_on_host, float *C_host);
int  clm4_sharded_gemm_result(const clm4_shard_ctx *ctx, int part, const float **C_dev);
/* The same product as a loop, with the C ROW PANELS ALL-GATHERED: every device ends with the whole C (rows x N fp32).
 *   clm4_sharded_gemm_begin    B to every device (stream-ordered), two full C buffers per device, `slots` timed steps;
 *   clm4_sharded_gemm_enqueue  step i: every device's clm4_gemm into its panel of C buffer i & 1 on its compute stream, then ONE in-place
 *                              ncclAllGather of that buffer per device on its exchange stream (ragged shards: a broadcast per owner;
 *                              shards repeating a device: copies) -- overlapping the kernel of step i + 1; no host wait;
 *   clm4_sharded_sync / clm4_sharded_step_timing (kernel ms, exchange ms) as for the mvm loop;
 *   clm4_sharded_gemm_full     device pointer of the whole C in buffer `buf` on shard `part`.
 * Bit-identical to clm4_gemm on the unsharded matrix.  A context runs ONE loop at a time (mvm or gemm: they share the step events). */
int  clm4_sharded_gemm_begin(clm4_shard_ctx *ctx, const int8_t *B, const float *sB, uint64_t N, int b_on_host, int slots);
/* The same loop with a choice of what is exchanged after a step's kernels (the all-gather moves (N - 1) / N of C into EVERY device --
 * 224 MiB per device and step at configs[3] on 8 GPUs against 0.06 ms of kernel: exchange-bound by construction):
 *   CLM4_GEMM_ALL_GATHER   = clm4_sharded_gemm_begin: every device ends with the whole C;
 *   CLM4_GEMM_GATHER_ROOT  the panels go to shard 0's device only (grouped ncclSend / ncclRecv): clm4_sharded_gemm_full(part 0) is the whole C,
 *                          the other devices' buffers hold their own panel;
 *   CLM4_GEMM_SHARDED      nothing is exchanged, C stays row-sharded like A: clm4_sharded_gemm_full(part) + row_begin(part) * N is shard
 *                          `part`'s panel.  The step is the kernel.
 * The bits of C never depend on the mode. */
#define CLM4_GEMM_ALL_GATHER 0
#define CLM4_GEMM_GATHER_ROOT 1
#define CLM4_GEMM_SHARDED 2
int  clm4_sharded_gemm_begin_mode(clm4_shard_ctx *ctx, const int8_t *B, const float *sB, uint64_t N, int b_on_host, int slots, int mode);
int  clm4_sharded_gemm_enqueue(clm4_shard_ctx *ctx, int step, int timed);
int  clm4_sharded_gemm_full(const clm4_shard_ctx *ctx, int part, int buf, const float **C_dev);

/* ---- synthetic data (bench / tests): fills device buffers without an fp32 source ------------------ */
/* nibbles uniform in [-7,7], scales uniform in [0.5,2): counter-based splitmix64 of (seed, index), so any
 * row range of a sharded matrix regenerates independently. */
int  clv_fill_random_nibbles(int8_t *q, uint64_t bytes, uint64_t seed, uint64_t byte_offset, void *stream);
int  clv_fill_random_scales(float *s, uint64_t count, uint64_t seed, uint64_t index_offset, void *stream);
/* fp32 integers uniform in [-range, range] (the reference's setRandomInteger(10) data, 01_measure.h:797) */
int  clv_fill_random_ints_f32(float *x, uint64_t count, int range, uint64_t seed, uint64_t index_offset, void *stream);

#ifdef __cplusplus
}
#endif
#endif
