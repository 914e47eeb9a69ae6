/*
 * ctn_abi.h - C ABI of the MI355X (gfx950) tensor-network contraction engine.
 *
 * This is the drop-in boundary for ContracTN's stabilised pairwise-contraction
 * executor.  It replaces, for the one hot path:
 *
 *   reference contractn/einsum.py:326-393  _core_contract(operands, contract_list, backend)
 *   reference contractn/einsum.py:89-107   stabilize()   (fused into every step's epilogue)
 *   reference contractn/einsum.py:110-114  destabilize() (left to the host: one exp + multiply)
 *   reference contractn/einsum.py:313-323  _contract_path() result -> ctn_plan (cached handle)
 *
 * A maintainer binds it with ctypes from contractn/einsum.py:303-305 (see
 * INTEGRATION.md).  Strings never cross the boundary: einsum symbols are mapped
 * to int32 labels, opt_einsum's shrinking operand positions to SSA tensor ids
 * (inputs 0..n_inputs-1, step k defines id n_inputs+k; the operand popped from
 * the HIGHER position is the step's lhs, reference einsum.py:344).
 *
 * Plain C types only.  No entry point throws or aborts; each returns CTN_OK or
 * a negative ctn_status and leaves a message in ctn_last_error() (thread local).
 *
 * Threading: a ctn_plan is immutable after creation and may be shared between
 * threads.  A ctn_exec owns mutable device state (workspace, pointer table,
 * partial-sum buffers, stream) and must not be used by two threads at once;
 * distinct ctn_exec objects may run concurrently.
 */
#ifndef CTN_ABI_H
#define CTN_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTN_ABI_VERSION 5

typedef enum {
  CTN_OK = 0,
  CTN_INVALID_ARG = -1,    /* -> TypeError / AssertionError on the Python side */
  CTN_SHAPE_MISMATCH = -2, /* -> ValueError */
  CTN_UNSUPPORTED = -3,    /* -> NotImplementedError: a dtype other than f32 / f64; an index group (batch, rows, columns,
                              contracted) of 2^31 or more entries; a CONTRACTED group that spans 2^31 or more elements of
                              an operand (tensors of 2^31 elements and more are fine otherwise: batch offsets are 64-bit
                              and outer free labels become batch labels; only the contracted group of a step needs
                              32-bit offsets - the engine lays its own large intermediates out accordingly); a streaming
                              step with 2^31 or more work items */
  CTN_OOM = -4,            /* -> MemoryError */
  CTN_HIP_ERROR = -5,      /* -> RuntimeError */
  CTN_RCCL_ERROR = -6,     /* reserved: collectives are issued by the host layer */
  CTN_NO_DEVICE = -7       /* -> RuntimeError: no gfx950 device visible */
} ctn_status;

typedef enum { CTN_F32 = 0, CTN_F64 = 1 } ctn_dtype;

/* where the caller's tensors live */
typedef enum { CTN_MEM_HOST = 0, CTN_MEM_DEVICE = 1 } ctn_memspace;

/* which kernel family a step was lowered to (ctn_step_info.kernel) */
typedef enum {
  CTN_KERNEL_ELEMENT = 0, /* streaming gather-multiply: one thread per 16-byte output vector, K loop
                             (copy-tensor / hyperedge products, Khatri-Rao, traces, small-K steps); one output per
                             thread with 16-byte loads along a short unit-stride K; K split over workgroups
                             (slabs + fixed-order reduce) when the outputs are few and K is long */
  CTN_KERNEL_DOT = 1,     /* one workgroup per output element, K split over lanes - and over workgroups from
                             K = 32768 on (slabs + fixed-order reduce) */
  CTN_KERNEL_MFMA_F32 = 2,/* 128x128 / 128x64 LDS-tiled v_mfma_f32_32x32x2_f32 GEMM, gather loads;
                             256x128 tiles fed by LDS-DMA when every tile is full (tile_m = 256);
                             64x64 split-K form when a launch cannot fill the chip */
  CTN_KERNEL_MFMA_F64 = 3,/* 128x64 LDS-tiled v_mfma_f64_16x16x4_f64 GEMM, gather loads; 128x128 tiles fed by
                             LDS-DMA (tile_n = 128); 64x64 split-K form for small launches */
  CTN_KERNEL_ROWDOT = 4,  /* one wave per output element, lanes along a unit-stride K (GEMV-like); K split over
                             workgroups as well when the outputs are few */
  CTN_KERNEL_FUSED = 5    /* no kernel: the step's result is formed on the fly inside the step that consumes it - an
                             element-wise (Khatri-Rao / Hadamard / broadcast) product as the A operand of an MFMA GEMM
                             (that step reports mode_a >= 3), or a GEMM whose small re-weighting consumer was regrouped
                             into it; the intermediate never exists and its rescale is reported as 0.0 */
} ctn_kernel_kind;

typedef struct ctn_plan ctn_plan;
typedef struct ctn_exec ctn_exec;

/*
 * Description of one contraction DAG = what `_contract_path` returns in the
 * reference (einsum.py:313-323) plus operand shapes, in integer form.
 */
typedef struct {
  int32_t dtype;                  /* ctn_dtype */
  int32_t n_inputs;
  const int32_t* in_ndim;         /* [n_inputs] */
  const int64_t* in_dims;         /* [sum in_ndim] extents, operand by operand */
  const int32_t* in_labels;       /* [sum in_ndim] integer label of every axis */
  const int64_t* in_strides;      /* [sum in_ndim] element strides, or NULL = C-contiguous */
  int32_t n_steps;                /* >= 1 */
  const int32_t* step_lhs;        /* [n_steps] SSA id of the left operand */
  const int32_t* step_rhs;        /* [n_steps] SSA id of the right operand, -1 = unary step */
  const int32_t* step_out_ndim;   /* [n_steps] */
  const int32_t* step_out_labels; /* [sum step_out_ndim]; axis order is honoured for the LAST
                                     step only - intermediates are laid out by the engine */
  int32_t stabilize;              /* 1 = rescale after every step (reference einsum.py:387); bit 1 (value 2) set in
                                     addition: the engine also chooses the axis order of the LAST step's result (its
                                     natural [batch][rows][columns] layout, read back with ctn_plan_out_labels) - for
                                     results that only feed another plan of the same caller (stages of a sliced
                                     contraction): the reference's results always have the caller's order */
  double min_norm;                /* 1e-7 in the reference (einsum.py:94) */
} ctn_plan_desc;

typedef struct {
  int32_t kernel;      /* ctn_kernel_kind */
  int32_t swapped;     /* 1 if the engine exchanged lhs/rhs so that the output's unit-stride label is a column label */
  int64_t batch;       /* |B|: labels shared by both operands and kept (hyperedge / batch) */
  int64_t m, n, k;     /* |M|, |N| free extents, |K| contracted extent (incl. summed-out labels) */
  int32_t mode_a;      /* 0 gather, 1 vector loads along the free index, 2 vector loads along k,
                          3 / 4 / 5 element-wise product of two tensors formed while the tile is staged (fused
                          step): scalar gathers / 16-byte accesses along the rows / along k */
  int32_t mode_b;
  int32_t partials;    /* abs-sum partials per replica written by the step (<= 512; more workgroups are collapsed to 1) */
  int32_t blocks;      /* workgroups per replica */
  double flops;        /* 2*|B||M||N||K| (|B||M||N| when K is empty) + 3*numel(out) */
  int64_t out_numel;
  int32_t tile_m;      /* MFMA steps: workgroup tile rows (fp32: 128, or 256 = LDS-DMA large-tile kernel); else 0 */
  int32_t tile_n;      /* MFMA steps: workgroup tile columns (fp32: 128 / 64, fp64: 64); else 0 */
  int32_t epilogue_sum; /* 2 / 4: the step's GEMM keeps one more (innermost) column label of this extent, which a third
                           tensor - a network input - re-weights and sums on the accumulator tile (an absorbed
                           `bpr,bp->br` after `bl,plr->bpr`: n counts the label, out_numel does not); else 0 */
  int32_t reserved;
} ctn_step_info;

/* ---- library ---------------------------------------------------------- */
int ctn_version(void);
const char* ctn_last_error(void);
/* number of visible HIP devices; CTN_NO_DEVICE if the runtime cannot be initialised */
int ctn_device_count(int* count);

/* ---- plan: pure host object, no HIP calls ----------------------------- */
int ctn_plan_create(const ctn_plan_desc* desc, ctn_plan** out);
void ctn_plan_destroy(ctn_plan* plan);
int ctn_plan_dtype(const ctn_plan* plan);
int ctn_plan_n_inputs(const ctn_plan* plan);
int ctn_plan_n_steps(const ctn_plan* plan);
/* algorithmic work of one contraction: sum over steps of ctn_step_info.flops */
double ctn_plan_flops(const ctn_plan* plan);
/* inputs read once + final output written once, in bytes */
int64_t ctn_plan_bytes_min(const ctn_plan* plan);
int ctn_plan_out_ndim(const ctn_plan* plan);
int ctn_plan_out_dims(const ctn_plan* plan, int64_t* dims /* [out_ndim] */);
/* labels of the result's axes, in order (the caller's own unless the plan was created with the free-order bit) */
int ctn_plan_out_labels(const ctn_plan* plan, int32_t* labels /* [out_ndim] */);
int64_t ctn_plan_out_numel(const ctn_plan* plan);
int64_t ctn_plan_out_bytes(const ctn_plan* plan);
/* device bytes an executor for `replicas` simultaneous contractions will allocate */
int64_t ctn_plan_workspace_bytes(const ctn_plan* plan, int replicas);
int ctn_plan_step_info(const ctn_plan* plan, int step, ctn_step_info* info);

/* ---- executor: device state for `replicas` independent contractions --- */
/* stream: a hipStream_t, or NULL to let the executor create its own */
int ctn_exec_create(const ctn_plan* plan, int device, void* stream, int replicas, ctn_exec** out);
void ctn_exec_destroy(ctn_exec* exec);

/*
 * Synchronous contraction of `replicas` networks.
 *   inputs  [replicas * n_inputs] operand pointers, replica-major; borrowed, never written
 *   outs    [replicas] caller-allocated, ctn_plan_out_bytes() each, C-contiguous
 *   log_scale      host [replicas]: sum over steps of log(rescale), accumulated on the device
 *   step_rescales  host [replicas * n_steps] or NULL: the rescale factor of every step
 *                  (0.0 where the step was not rescaled) so the caller can redo the
 *                  log accumulation in the reference's exact order and precision.
 *                  The PRODUCT of the factors (the register, and with it (T_hat, c)) is the reference's; the
 *                  factor of an individual step is the reference's own (what cpu_ref.core_contract(record=True)
 *                  lists, einsum.py:97-106) only where the step ran as a launch of its own.  A step whose result
 *                  never exists in memory reports 0.0 and the NEXT launched step carries the product of both:
 *                    - ctn_step_info.kernel == CTN_KERNEL_FUSED (formed inside its consumer);
 *                    - the first step of a zipper pair run by k_zip_f32 (ctn_exec_step_tile reports (1, 1) for
 *                      it): (T / s) Y = (T Y) / s, the second step's factor is s_T * s_E' of the reference;
 *                  the members of a sweep (k_sweep_f32, tile (1, 1) as well) DO report the reference's per-step
 *                  factors, reconstructed after the launch (within 2e-5 relative, fp32).
 */
int ctn_exec_run(ctn_exec* exec, const void* const* inputs, int inputs_space,
                 void* const* outs, int outs_space, double* log_scale, double* step_rescales);

/* Asynchronous variant: device pointers only, returns after enqueueing on the stream. */
int ctn_exec_enqueue(ctn_exec* exec, const void* const* dev_inputs, void* const* dev_outs);
/* Wait for the stream and copy out the scale registers of the last enqueue (either may be NULL).
 * The operands and outputs of that enqueue stay borrowed until this call returns: when the scale registers
 * show that a lazily rescaled product left the dtype's range (see ctn_exec_set_rescale_mode) the contraction is
 * repeated here, into the same output buffers, before anything is reported. */
int ctn_exec_fetch(ctn_exec* exec, double* log_scale, double* step_rescales);
int ctn_exec_synchronize(ctn_exec* exec);

/*
 * Where stabilize() (reference einsum.py:89-107, called at :387 after every step) is applied.
 *   0  lazy, the default: a step stores its un-normalised output and consumers fold 1 / (sA sB) into their
 *      epilogue - no extra pass over HBM.  The tile kernels then accumulate on un-normalised operands, which
 *      can leave the dtype's range where the reference stays finite (fp32 operands of magnitude ~1e13); every
 *      fetch checks the scale registers for that and, if so, switches the executor to mode 1 and repeats the
 *      contraction (ctn_exec_eager_reruns counts how often).
 *   1  eager: every intermediate is divided by its rescale in place right after its step and consumed with
 *      scale 1 - the reference's own order of operations, one extra read+write per intermediate.
 * Returns the previous mode, or a negative ctn_status.
 */
int ctn_exec_set_rescale_mode(ctn_exec* exec, int mode);
int ctn_exec_eager_reruns(const ctn_exec* exec);

/*
 * destabilize() (reference einsum.py:110-114: T_hat * exp(log_scale), what contract(split_format=False) returns) in the
 * same pass over the final tensor as the last stabilize() division, for results that stay on the device (a 4 GiB
 * result is then read and written once after its last step, not twice):
 *   ctn_exec_set_finish_mode(exec, 1): the runs that follow leave the final tensor UN-normalised (registers and
 *     per-step factors are produced as always); returns the previous mode.  0 restores the default.
 *   ctn_exec_finish(exec, mult): after ctn_exec_fetch of such a run, on the executor's stream:
 *     out_r = (out_r / rescale_last_r) * mult[r]  in the tensor dtype, both roundings as the reference's two
 *     operations make them; mult = host [replicas], normally exp(log_scale) in the register's precision.
 */
int ctn_exec_set_finish_mode(ctn_exec* exec, int mode);
int ctn_exec_finish(ctn_exec* exec, const double* mult);

/*
 * Device-side join of partial results in split format (the index slices of one network run as the
 * replicas of one executor, each producing (T_hat_s, c_s); ranks exchange ONE packed buffer).  The reference has
 * no counterpart (single process); the arithmetic is that of its stabilize(), einsum.py:89-107, applied to a sum.
 *
 * ctn_exec_snapshot_scales: after an enqueue, copy the log-scale registers of its first `n` replicas to
 *   `dev_log_dst` (device, n doubles) and - optionally - their per-step rescale factors to `host_rescales`
 *   (pinned host memory, n * n_steps doubles), both asynchronously on the executor's stream: no host wait.
 * ctn_exec_scales_suspect: the range check ctn_exec_fetch applies (see ctn_exec_set_rescale_mode), on rescale
 *   factors the caller has brought to the host itself; 1 = a lazily rescaled product may have left the dtype's
 *   range (repeat that enqueue and ctn_exec_fetch it), 0 = fine, negative = ctn_status.
 * ctn_exec_report_suspect: a caller that never fetches (it keeps everything on the device and checks the range itself
 *   with ctn_exec_scales_suspect) tells the executor the verdict of the run, once per run: consecutive suspect runs
 *   count like a fetch's own findings, and from the third on the executor stays in eager mode (whose runs are never
 *   suspect) instead of paying a lazy pass plus the caller's checked repeat every time.  Returns the streak.
 * ctn_exec_combine_split: on the executor's stream, out_packed[0 .. numel) = T_hat (as doubles) and
 *   out_packed[numel] = c of  sum_i t_i e^{c_i}  over n parts; part i is t + i * t_stride elements of `t_dtype`,
 *   its scale c[i * c_stride]; exact zeros are left out of the maximum, the sum is re-stabilised (mean |T_hat| = 1
 *   when sum |T| > min_norm of the plan), bit-reproducible.  n <= 4096; all pointers are device pointers.
 */
int ctn_exec_snapshot_scales(ctn_exec* exec, double* dev_log_dst, int n, double* host_rescales);
int ctn_exec_scales_suspect(const ctn_exec* exec, const double* host_rescales, int replicas);
int ctn_exec_report_suspect(ctn_exec* exec, int suspect);
int ctn_exec_combine_split(ctn_exec* exec, int t_dtype, const void* t, int64_t t_stride, const double* c,
                           int64_t c_stride, int n, int64_t numel, double* out_packed);

/*
 * Scale bookkeeping between the STAGES of a sliced contraction (several plans of one caller whose results feed each
 * other and never leave the device: a lower stage's evaluation is an operand of the stage above, its log-scale
 * register rides along).  Both run on the executor's stream; all pointers are device pointers unless noted.
 *
 * ctn_exec_add_scales: dst[i] = own[i] + sum_j kid_scales[j][kid_index[j][i]], i < n, added in the order of j
 *   (kid_scales / kid_index: HOST arrays of n_kids device pointers).  dst may be own.
 * ctn_exec_merge_scales: `n` evaluations of `numel` elements of `t_dtype`, `stride` elements apart (16-byte aligned),
 *   with registers scales[n], lie on a row-major grid extents[ndim] (ndim <= 8, product n <= 65535); along the axes
 *   with merged[d] != 0 they are about to be read as ONE strided operand, so every group of evaluations that differ
 *   only along those axes is multiplied up to its common scale: top = max register among the members that are not
 *   exact zeros (the whole tensor is looked at), member i *= e^{scales[i] - top}, scales[i] = top (0 for a group of
 *   zeros; exact zeros are not touched).
 */
int ctn_exec_add_scales(ctn_exec* exec, double* dst, const double* own, int n, int n_kids,
                        const double* const* kid_scales, const int64_t* const* kid_index);
int ctn_exec_merge_scales(ctn_exec* exec, int t_dtype, void* buf, int64_t stride, int64_t numel, double* scales, int n,
                          int ndim, const int32_t* extents, const int32_t* merged);

/*
 * Workgroup tile (rows, columns) of the MFMA kernel that the LAST enqueue launched for `step`
 * (0, 0 for non-MFMA steps or before the first enqueue).  The planner's choice (ctn_step_info.tile_m/n)
 * can be overridden at launch time by the number of replicas: few tiles -> 64 x 64 split-K or 128 x 64,
 * many full long-K tiles -> 256 x 256.  Measurement tooling only (bench.py keys its per-kernel roofline on it).
 */
int ctn_exec_step_tile(const ctn_exec* exec, int step, int32_t* tile_m, int32_t* tile_n);

/*
 * Per-step device timing with HIP events recorded on the executor's stream.
 * ctn_exec_set_timing(exec, slots): slots > 0 brackets every step's kernels of
 * the next `slots` enqueues with events (later enqueues run without events);
 * 0 switches timing off.  ctn_exec_step_ms() waits for the
 * stream and returns, per step, the mean duration over the recorded enqueues
 * (ms[n_steps]).
 */
int ctn_exec_set_timing(ctn_exec* exec, int slots);
int ctn_exec_step_ms(ctn_exec* exec, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* CTN_ABI_H */
