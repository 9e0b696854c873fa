/* dfft.h -- C-ABI of libdfft_mi355x.so: the MI355X-native slab 3D C2C FFT.
 *
 * This is the drop-in boundary for the hot path of lueelu/DistributedFFT's 3dmpifft_opt
 * (t0 batched 2D YZ FFT -> t1 pack -> t2 all-to-all -> t3 batched 1D X FFT).  Every entry point names the reference
 * interface it replaces (paths relative to /root/reference/3dmpifft_opt/include/).  The reference API has C++ linkage
 * (fft_mpi_3d_api.h:68-79); thin C++ wrappers with the original names live in include/fft_mpi_3d_api.h and call
 * straight into these functions, so the reference driver (fftSpeed3d_c2c.cpp) recompiles unchanged against this
 * library.  Plain pointers and sizes only: no torch, no MPI, no C++ types in any signature.
 *
 * Error handling: functions returning int return 0 on success and a negative DFFT_E* code on failure (the C++
 * wrappers reproduce the reference behaviour: print "[file:line] ... failed" and exit(EXIT_FAILURE),
 * fft_mpi_common.h:31-103).  dfft_last_error() returns a thread-local message for the last failure.
 *
 * Layout contract, device g of P, xl = ceil(N0/P), yl = ceil(N1/P), last device takes the rest:
 *   forward  input  : [x_local][N1][N2]  (x slowest)            element (xi*N1 + y)*N2 + z
 *   forward  output : [y_local][N2][N0]  (kx fastest)           element (yy*N2 + z)*N0 + kx   (transposed, Y-slabbed)
 *   backward input  : the forward output layout;  backward output: the forward input layout.  Both unnormalised.
 */
#ifndef DFFT_H
#define DFFT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFFT_FORWARD 1   /* fft_mpi_common.h:18  FORWARD  */
#define DFFT_BACKWARD (-1) /* fft_mpi_common.h:19  BACKWARD */
#define DFFT_ALLOC_HOST 1  /* fft_mpi_common.h:15  ALLOC_CPU */
#define DFFT_ALLOC_DEV (-1) /* fft_mpi_common.h:16  ALLOC_DEV */

#define DFFT_F64 0 /* Complex = double[2] (fft_mpi_common.h:21); the only precision the reference has */
#define DFFT_F32 1 /* float[2]; BASELINE config 5 */

/* error codes */
#define DFFT_OK 0
#define DFFT_EINVAL (-1)      /* bad argument / unsupported size */
#define DFFT_EHIP (-2)        /* a HIP runtime call failed */
#define DFFT_ERCCL (-3)       /* an RCCL call failed */
#define DFFT_ENOGPU (-4)      /* no usable gfx950 device: the product path has no CPU fallback */
#define DFFT_ECOMM (-5)       /* bootstrap / rendezvous failure */
#define DFFT_EUNSUPPORTED (-6)

/* plan flags */
#define DFFT_PLAN_DEFAULT 0u
#define DFFT_PLAN_UNFUSED 1u        /* reference stage structure: Y-FFT in place, separate pack (t1), separate tile transpose
                                       in t3.  Default is fused (pack and transpose folded into the FFT kernels' stores). */
#define DFFT_PLAN_INPUT_FROM_IN 2u  /* every execute re-reads the caller's `in` (first pass runs out-of-place in -> bufferDev1)
                                       instead of consuming bufferDev1; out-of-place plans only.  Same HBM traffic. */
#define DFFT_PLAN_OVERLAP 4u        /* P > 1: the exchange runs in pieces on a second stream -- X-plane parts overlapped with
                                       the YZ stage, Y sub-blocks with the X stage (forward plans; backward plans that also
                                       set DFFT_PLAN_INPUT_FROM_IN).  Results are bit-identical to the serial pipeline. */
#define DFFT_PLAN_NATURAL 8u        /* input AND output in the natural X-slab layout [x_local][N1][N2] (both directions):
                                       the un-transposed output the reference declares (fft_mpi_local_size_3d,
                                       fft_mpi_3d_api.h:73) but never implements.  Costs a second all-to-all when P > 1. */

/* execute flags */
#define DFFT_EXEC_ASYNC 0u          /* enqueue on the plan's stream and return */
#define DFFT_EXEC_SYNC_STAGES 1u    /* hipDeviceSynchronize-style host timing per stage, like fft_mpi_3d_api.cpp:184-201 */
#define DFFT_EXEC_PRINT 2u          /* print the reference's "t0: .. t1: .. t2: .. t3: .. total: .." line (forward only) */
#define DFFT_EXEC_NO_TIMING 4u      /* do not record the stage-boundary events (dfft_stage_times is then unavailable): a
                                       production loop of small transforms is launch-bound, and the 5-6 event records per
                                       execute are as many queue packets as the kernels themselves */

typedef struct dfft_plan_s* dfft_plan_t;
typedef struct dfft_comm_s* dfft_comm_t;

/* ---- library / device ------------------------------------------------------------------------------------------ */
const char* dfft_version(void);
const char* dfft_last_error(void);
/* Number of visible HIP devices (0 if none).  hipGetDeviceCount in fftSpeed3d_c2c.cpp:33-34. */
int dfft_device_count(void);
/* PCI address ("domain:bus:device.function", hipDeviceGetPCIBusId) of HIP device `device` (-1: the calling thread's current
 * device): the identity a multi-process launch compares to prove that its ranks sit on distinct GPUs -- ordinals do not
 * (HIP_VISIBLE_DEVICES renumbers them per process).  Returns 0 and a NUL-terminated string in buf[0..len). */
int dfft_device_pci_bus_id(int device, char* buf, int len);
/* 1 if FFT length n is supported: any product of 2, 3, 5, 7 up to 4096 (tuned plans for the lengths listed in
 * csrc/dfft_plans.h, a run-time-scheduled kernel for the rest) -- the single-pass range of the reference's generator --
 * and, above 4096, every product of two tuned lengths up to 2^24 (two-pass "four-step" plans, csrc/dfft_long.hip; the
 * reference's multi-upload plans, templateFFT.cpp:3972-4106).  3D plans with such an axis run the un-fused stage structure. */
int dfft_length_supported(long long n);

/* ---- slab bookkeeping: pure host arithmetic, callable without a GPU ------------------------------------------------ */
/* getProperDeviceNum (fft_mpi_3d_api.cpp:232-272): shrink the device count when N0 % P != 0 so every device but the
 * last owns ceil(N0/P) planes.  real_devices < 0 skips the clamp to the visible device count. */
int dfft_proper_device_count(const long long N[3], int ini_devices_in_rank, int nranks, int rank, int real_devices,
                             int* new_total, int* new_in_rank);
/* getDataCountForNode (fft_mpi_3d_api.cpp:274-287): elements held by global device idx before the transform. */
long long dfft_local_count(const long long N[3], int total_devices, int global_idx);
/* getMaxDataCount (fft_mpi_3d_api.cpp:289-316): elements each of in/out/bufferDev1 must hold. */
long long dfft_max_count(long long n0, long long n1, long long n2, int total_devices, int is_last_device);
/* Per-peer exchange counts/offsets in elements (tInfo, fft_mpi_3d_api.cpp:84-133; receive offsets :618-625).
 * Arrays have total_devices entries.  direction = DFFT_FORWARD or DFFT_BACKWARD. */
int dfft_exchange_layout(long long n0, long long n1, long long n2, int total_devices, int global_idx, int direction,
                         long long* scount, long long* soffset, long long* rcount, long long* roffset);
/* The messages of ONE piece of the overlapped exchange (DFFT_PLAN_OVERLAP) as device global_idx issues them: X-plane part
 * `part` when every device cuts its X slab into parts of `part_planes` planes, restricted to Y sub-block `ycut` of `ycuts`
 * (or all sub-blocks for ycut = -1).  Message m goes to / comes from peer[m]; offsets/counts in elements.
 *   forward : send buffer packed [k][dst][x][y in k][N2] ([dst][x][y][N2] for ycuts = 1), receive buffer [k][x][y in k][N2]
 *   backward: the mirror image -- send buffer [k][x][y in k][N2], receive buffer [k][src][x][y in k][N2]
 * Both ends enumerate the messages of a pair in the same order.  Returns the number of messages, or a negative error.
 * (Refines slabAlltoall's per-peer chunks, fft_mpi_3d_api.cpp:610-672, into the pieces the pipeline overlaps.) */
int dfft_exchange_part_layout(long long n0, long long n1, long long n2, int total_devices, int global_idx, int direction,
                              long long part_planes, int part, int ycuts, int ycut, int max_msgs, int* peer,
                              long long* soffset, long long* scount, long long* roffset, long long* rcount);
/* local extents: x planes owned before / y rows owned after the forward transform, and their global starts.
 * (the declared-but-never-defined fft_mpi_local_size_3d, fft_mpi_3d_api.h:73) */
int dfft_local_size(long long n0, long long n1, long long n2, int total_devices, int global_idx, long long* local_n0,
                    long long* local_0_start, long long* local_n1, long long* local_1_start);

/* ---- exchange communicators (t2) -------------------------------------------------------------------------------------
 * LOCAL : all P devices are driven by threads of this process (reference: OpenMP thread per GPU + hipMemcpyPeerAsync,
 *         fft_mpi_3d_api.cpp:613-630).  Also serves P "virtual" devices sharing one physical GPU (parity tests).
 * RCCL  : one communicator rank per device, any process layout; replaces the MPI_Isend/Irecv on device pointers
 *         (fft_mpi_3d_api.cpp:635-672) with grouped ncclSend/ncclRecv over xGMI. */
int dfft_comm_create_local(int total_devices, dfft_comm_t* comm);
/* 128-byte RCCL unique id; rank 0 creates it and the host distributes it (torch.distributed, MPI, dfft_boot_*). */
int dfft_rccl_unique_id(char id[128]);
int dfft_comm_create_rccl(const char id[128], int total_devices, int global_idx, dfft_comm_t* comm);
/* IPC   : one process per device WITHOUT RCCL: the receive buffers are shared through hipIpc handles (exchanged over the
 *         dfft_boot_* rendezvous, which must describe exactly these processes), peers push their chunks with device-to-device
 *         copies (SDMA engines, no CUs), and the processes synchronise through the rendezvous' barrier -- the cross-process
 *         twin of the LOCAL communicator, i.e. the reference's MPI path with hipMemcpy instead of UCX (fft_mpi_3d_api.cpp:
 *         635-672).  Host-synchronising, so slower than RCCL for small messages; it also works with several processes sharing
 *         one GPU, which is how the multi-process path is tested on a single-GPU machine.  Plan creation and destruction are
 *         collective over the processes.  async_exchange != 0: the barriers become flag words in IPC-shared fine-grained
 *         memory, published and awaited by one-wave kernels on the plan's stream, so the exchange is stream-ordered like
 *         RCCL's (nothing blocks the host, DFFT_PLAN_OVERLAP overlaps) while the data still moves by copy engines. */
int dfft_comm_create_ipc(int total_devices, int global_idx, int async_exchange, dfft_comm_t* comm);
/* What a communicator actually is, as the transport reports it (launch diagnostics: a multi-GPU benchmark must be able to
 * prove which back-end ran and over how many ranks).  kind: 0 LOCAL, 1 RCCL, 2 IPC (host-synchronised), 3 IPC (stream-ordered);
 * size/rank: for RCCL the values of ncclCommCount / ncclCommUserRank (not the arguments the caller passed), otherwise the
 * creation arguments; device: HIP device ordinal the communicator is bound to (ncclCommCuDevice for RCCL, -1 for LOCAL).
 * Any output pointer may be NULL.  (The reference has no counterpart: MPI_Comm_size/rank, fftSpeed3d_c2c.cpp:20-21.) */
int dfft_comm_info(dfft_comm_t comm, int* kind, int* size, int* rank, int* device);
int dfft_comm_destroy(dfft_comm_t comm);

/* ---- memory ------------------------------------------------------------------------------------------------------------
 * fft_mpi_alloc_local_memory (fft_mpi_3d_api.cpp:216-230); count in complex elements of dtype. */
void* dfft_alloc(long long count, int dtype, int flag);
int dfft_free(void* p, int flag);

/* ---- plan / execute ------------------------------------------------------------------------------------------------------
 * fft_mpi_plan_dft_c2c_3d (fft_mpi_3d_api.cpp:41-141).  `in`/`out` are device buffers of dfft_max_count elements owned
 * by the caller; out == NULL or out == in selects in-place (bufferDev2 = in).  The plan allocates bufferDev1 and copies
 * `in` into it (input is captured at plan time or by writing dfft_plan_buffer1()).  comm may be NULL when
 * total_devices == 1.  The calling thread's current HIP device is the plan's device. */
int dfft_plan_create(dfft_plan_t* plan, long long n0, long long n1, long long n2, int dtype, int direction, void* in,
                     void* out, dfft_comm_t comm, int global_idx, int total_devices, unsigned flags);
/* plan->bufferDev1, which the reference driver writes directly (fftSpeed3d_c2c.cpp:78). */
void* dfft_plan_buffer1(dfft_plan_t plan);
/* the buffer holding the result after execute (bufferDev2 = out, or in when in-place). */
void* dfft_plan_result(dfft_plan_t plan);
void* dfft_plan_stream(dfft_plan_t plan); /* hipStream_t the plan enqueues on */
/* Diagnostics: the plan's internal hand-over buffer between the passes (NULL when the plan has none) and its size in bytes.
 * No counterpart in the reference (its intermediate is bufferDev1 itself, fft_mpi_3d_api.cpp:497). */
void* dfft_plan_workbuf(dfft_plan_t plan, long long* bytes);
/* fft_mpi_execute_dft_3d_c2c (fft_mpi_3d_api.cpp:181-214).  Collective over all devices of the communicator. */
int dfft_execute(dfft_plan_t plan, unsigned exec_flags);
/* Wait for the plan's stream. */
int dfft_plan_sync(dfft_plan_t plan);
/* Optional plan-time measurement (the FFTW_MEASURE of this library; no counterpart in the reference).  The X pass of a
 * single-GPU plan reads the plan's internal hand-over buffer and writes the result buffer; it runs 5-8 % faster when the two
 * lie in different regions of the device's physical memory (profiles/r03/README.md section 1), which consecutive allocations
 * usually do not.  dfft_plan_tune times that ONE kernel -- seven launches of ~0.7 ms per candidate, no complete transforms --
 * on the current buffer and on fresh allocations of the same size made one after the other and all kept until the end (memory
 * is handed out in runs of 2 ... 36 such allocations that behave alike, so a dense walk cannot step over a run), stops as soon
 * as two candidates differ by 3 % and a second timing of the fastest and the slowest confirms a 3.5 % gap, keeps the fastest and frees
 * everything else.  Bounds: DFFT_TUNE_TRIES candidates and a transient footprint of DFFT_TUNE_MEM_PCT per cent of the FREE device
 * memory -- by default 32 candidates / 25 %, because other plans and processes may share the GPU, and 128 / 70 % when at least 90 % of
 * the device's memory is free, i.e. when this process evidently has the GPU to itself (on some boxes 60+ GiB of consecutive
 * allocations behave alike and the short walk finds no fast buffer: profiles/r04/experiments/tune_check_short_walk.log); worst case
 * about a second at 512^3 fp64.  The probe launches overwrite the result buffer (forward plans) with garbage -- its contents are set
 * aside and put back:
 * call it before the first execute, not between an execute and the use of its result.  A no-op for plans without such a
 * buffer (P > 1, un-fused, natural-order, cache-resident sizes) and with DFFT_TUNE=0.  Results of later executes are
 * bit-identical with and without tuning.  The reference-named wrapper fft_mpi_plan_dft_c2c_3d, distFFTOpt, speed3d_c2c and
 * bench.py all call it for out-of-place plans, so the drop-in CLI times the same configuration as the benchmark. */
int dfft_plan_tune(dfft_plan_t plan);
/* One line of text about how this plan executes: pipeline (fused / unfused / natural), whether the YZ stage is one persistent
 * launch or two launches per cache chunk, the chunk geometry, where the intermediate lives, whether the exchange buffers' rows
 * are rotated, the overlap geometry, and whether dfft_plan_tune has placed the hand-over buffer.  Diagnostics (bench.py puts it
 * in its JSON line); no counterpart in the reference.  buf must hold at least 64 bytes. */
int dfft_plan_describe(dfft_plan_t plan, char* buf, int len);
/* What the last dfft_plan_tune of this plan saw: ms[i] = X-pass kernel time on candidate i (at most max_n are written), *kept =
 * index of the candidate the plan now uses (-1: never tuned), *final_ms = the kept candidate re-timed after the others were
 * freed.  Returns the number of candidates tried.  Any output pointer may be NULL. */
int dfft_plan_tune_report(dfft_plan_t plan, int max_n, double* ms, int* kept, double* final_ms);
/* Multiply the result of every later execute by s (e.g. 1/N for a normalised transform: heFFTe's scale::full, the
 * reference's scale_element pass in 3dmpifft_roc, kernel_func.cpp:102-157).  Folded into the X-pass kernel's store, so it
 * costs no extra pass over the data.  s = 1 (the default) reproduces the reference's un-normalised transforms. */
int dfft_plan_set_scale(dfft_plan_t plan, double s);
/* Stage times of the last forward/backward execute in seconds: t[0..3] = t0..t3 (backward: X, exchange, unpack, YZ),
 * from HIP events on the plan's stream (ASYNC) or host clocks (SYNC_STAGES).  Syncs the stream. */
int dfft_stage_times(dfft_plan_t plan, double t[4]);
/* Durations in seconds of the three FFT kernels of the last ASYNC execute of a fused plan: t[0] = Z rows, t[1] = Y columns
 * (+pack), t[2] = X columns (+transpose); HIP events on the plan's stream.  Used for the roofline figures. */
int dfft_kernel_times(dfft_plan_t plan, double t[3]);
/* fft_mpi_destroy_plan (fft_mpi_3d_api.cpp:143-179). */
int dfft_plan_destroy(dfft_plan_t plan);

/* ---- batched 1D building block (the kernels behind t0/t3; templateFFT batchTest-style checks) ---------------------------
 * In-place or out-of-place length-n C2C FFT of `batch` contiguous rows (stride n). */
int dfft_fft1d_rows(void* in, void* out, long long n, long long batch, int dtype, int direction, void* stream);
/* Length-n FFT down the columns of a [n][width] row-major matrix, `batch` matrices back to back. */
int dfft_fft1d_cols(void* in, void* out, long long n, long long width, long long batch, int dtype, int direction,
                    void* stream);

/* ---- batched 2D transform (templateFFT's FFTDim = 2 application: initializeFFT, templateFFT.cpp:5767, launched by fftZY,
 * fft_mpi_3d_api.cpp:466-522; component benchmark templateFFT/batchTest/Test_2D.cpp:29-198) ---------------------------------
 * `batch` planes of [n1][n2] complex elements (n2 contiguous), each transformed along both axes, in place (out == in) or out
 * of place (`in` is left untouched).  This is the t0 stage of a 3D plan as an entry point of its own: planes are processed in
 * groups that fit the 256 MiB Infinity Cache (the column pass reads what the row pass wrote from the cache), and the plane
 * shapes the one-launch stage is built for (fp64; n1 = 256 / 512 with n2 = 256 / 512, n1 = 768 with n2 = 512) run as ONE
 * persistent launch per call.  Un-normalised in both directions.  The one-launch form keeps a small control block per (device,
 * stream, plane shape, direction), freed by dfft_trim(); a launch that gives up (see dfft_zy.hip) is reported by the next call
 * on that stream, which -- like every later one -- runs on two launches per chunk. */
int dfft_fft2d_batch(void* in, void* out, long long n1, long long n2, long long batch, int dtype, int direction, void* stream);

/* Frees the scratch buffers the 1-D entry points cache per (device, stream) for lengths above 4096 (four-step transforms) and the
 * control blocks of dfft_fft2d_batch.
 * Buffers in use by a call in progress are left alone.  No counterpart in the reference. */
int dfft_trim(void);

/* data[i] *= s for `count` complex elements on the device (the 1/N normalisation both transforms leave to the caller;
 * the reference's scale_element kernel, kernel_func.cpp:102-157, used only by 3dmpifft_roc). */
int dfft_scale(void* data, long long count, int dtype, double s, void* stream);

/* ---- tiny TCP rendezvous for multi-process launches without MPI ----------------------------------------------------------
 * Replaces what the reference driver needs from MPI besides moving data (MPI_Comm_rank/size, MPI_Bcast of the RCCL id,
 * MPI_Barrier, MPI_Reduce(MAX), fftSpeed3d_c2c.cpp:18-26,120-124).  Rank/size/address come from the environment:
 * DFFT_RANK/DFFT_WORLD_SIZE/DFFT_MASTER_ADDR/DFFT_MASTER_PORT, else torchrun's RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT,
 * else a PMI/OpenMPI launcher's PMI_RANK/PMI_SIZE or OMPI_COMM_WORLD_RANK/SIZE; single process if none is set. */
int dfft_boot_init(void);
int dfft_boot_rank(void);
int dfft_boot_size(void);
int dfft_boot_bcast(void* buf, size_t bytes, int root);
int dfft_boot_barrier(void);
int dfft_boot_allreduce_max(double* v, int n);
int dfft_boot_finalize(void);
/* Every wait of the rendezvous is bounded by DFFT_BOOT_TIMEOUT_S (default 180 s, 0 = unbounded): a peer that died or left the
 * collective call sequence yields DFFT_ECOMM naming the collective, the rank waited for and the reason.
 * Diagnostics (no counterpart in the reference, whose MPI calls simply hang): the library keeps the last 256 control-plane
 * events of the process (rendezvous collectives, buffer registrations, exchange rounds, RCCL calls, executes of P > 1 plans).
 * They are printed to stderr with every DFFT_ECOMM / DFFT_ERCCL failure (DFFT_TRACE_ON_ERROR=0: not), on SIGUSR2 together with
 * a native backtrace when DFFT_TRACE_SIGNAL=1 is set, and by this call. */
int dfft_trace_dump(void);

#ifdef __cplusplus
}
#endif
#endif /* DFFT_H */
