/*
 * hmm_engine.h — C ABI of the MI355X (gfx950) HMM forward / backward / posterior /
 * Viterbi engine.  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless stated otherwise; all calls are asynchronous on `stream` (a hipStream_t passed
 * as void*), re-entrant, and keep no state between calls (the only process-wide settings are
 * the explicit tuning options of hmm_set_option).
 *
 * The reference (sukui-genomics-cn/hmm_layer) has no native boundary: its hot path is a
 * Python loop over HmmCell.forward.  Each entry point below replaces one reference
 * driver; the file:line it replaces is cited per function.  Shapes follow the
 * reference: k = number of models, b = batch, L = sequence length, q = states,
 *   A   (k,q,q)   row-stochastic transition matrices   (transitioner.make_A(),
 *                 hmm_layer/gene_pred_hmm_transitioner.py:99-102)
 *   pi  (k,q)     start distributions                  (make_initial_distribution(), :111-112)
 *   E   (k,b,L,q) emission PROBABILITIES, row-major    (cell.emission_probs(),
 *                 hmm_layer/MsaHmmCell.py:61-71) — the engine applies max(.,eps) itself,
 *                 like the cell does (hmm_layer/MsaHmmCell.py:87-88).
 * All tensors are fp32 and contiguous; log-likelihoods are returned in fp64.
 *
 * Workspace: the caller owns all memory.  Query hmm_workspace_bytes() and pass a
 * device buffer of at least that size (256-byte aligned) to the call.
 *
 * Streams: to the caller every call is an ordinary in-order operation on `stream` — it starts
 * after everything enqueued there before it, and everything enqueued after it sees its results.
 * Inside, three calls put independent parts of their work on a per-device helper stream, forked
 * from and joined back into `stream` with events: hmm_posterior for q > 64 (the two recursions),
 * hmm_viterbi on large batches (batch groups, HMM_OPT_VGROUPS) and, opt-in, hmm_posterior
 * (HMM_OPT_GROUPS).  They remain capturable into a HIP graph and give identical results when no
 * helper stream can be created (everything then runs in order on `stream`).
 */
#ifndef HMM_ENGINE_H
#define HMM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMM_ENGINE_ABI_VERSION 3

/* error codes (0 = success); hmm_strerror() names them */
#define HMM_OK                 0
#define HMM_ERR_BAD_SHAPE     -1   /* k,b,L,q < 1 */
#define HMM_ERR_Q_UNSUPPORTED -2   /* q above what this build's kernels cover */
#define HMM_ERR_NULL_POINTER  -3
#define HMM_ERR_WORKSPACE     -4   /* workspace too small or misaligned */
#define HMM_ERR_LAUNCH        -5   /* HIP launch error (see hipGetLastError) */
#define HMM_ERR_BAD_ARGUMENT  -6
#define HMM_ERR_NO_DEVICE     -7
#define HMM_ERR_NO_RCCL       -8   /* hmm_loglik_allreduce: the process has no RCCL loaded */

/* operations, for hmm_workspace_bytes() */
#define HMM_OP_LOGLIK     0   /* hmm_forward without log_alpha */
#define HMM_OP_FORWARD    1   /* hmm_forward with log_alpha      */
#define HMM_OP_BACKWARD   2
#define HMM_OP_POSTERIOR  3
#define HMM_OP_VITERBI    4

/* output modes of hmm_posterior() */
#define HMM_POST_PROB        0   /* gamma (probabilities, rows sum to 1)                    */
#define HMM_POST_LOG         1   /* log gamma = log alpha + log beta - loglik               */
#define HMM_POST_LOG_NO_LL   2   /* log alpha + log beta (the reference's no_loglik=True)   */

const char *hmm_strerror(int code);
int hmm_abi_version(void);

/*
 * Tuning / test options: explicit, process-wide, read at launch time.  The defaults are the
 * measured best and results depend on nothing outside a call's arguments and these options; the
 * HMM_ENGINE_CHUNK / _FORCE_DENSE / _SCAN2 / _GROUPS / _EXACT environment variables only seed
 * them once, when the first option is read.  hmm_set_option returns the previous value
 * (HMM_ERR_BAD_ARGUMENT for an unknown option).
 */
#define HMM_OPT_CHUNK        0   /* chunk length of the scan (multiple of 16, <= 512); 0 = chosen per shape */
#define HMM_OPT_FORCE_DENSE  1   /* 1: generic kernels even for the compiled gene topologies               */
#define HMM_OPT_SCAN2        2   /* 0: single-level chunk scan                                              */
#define HMM_OPT_GROUPS       3   /* batch groups pipelined on two internal streams (1 = off)               */
#define HMM_OPT_EXACT        4   /* HMM_EXACT_*: routing to the serial exact-clamp kernels (q <= 16)        */
#define HMM_OPT_PGCHUNK      5   /* hmm_posterior_grad in chunks: 0 never, 1 when it pays (default), 2 always  */
#define HMM_OPT_VGROUPS      6   /* hmm_viterbi: batch groups pipelined on an internal stream; 0 = chosen per shape, 1 = off */
#define HMM_OPT_COUNT        7
#define HMM_EXACT_AUTO    0      /* decided on the device (see hmm_posterior)                               */
#define HMM_EXACT_OFF     1      /* always the chunked scan                                                 */
#define HMM_EXACT_ALWAYS  2      /* always the serial kernels                                               */
#define HMM_EXACT_ALWAYS_NARROW 3 /* test hook: as ALWAYS, with the one-sequence-per-wave layout that sequences
                                    of more than 2 GB / 16 need                                              */
int hmm_set_option(int option, int value);
int hmm_get_option(int option);

/* Largest q supported (4096): q <= hmm_scan_max_states() (16) runs the chunked scan kernels,
 * larger models run serial in time with one f32-MFMA GEMM per position (the profile-HMM sizes,
 * e.g. q = 2*512+3 = 1027); in between, up to 64 states: the chunked scan with 32- / 64-state tiles where it
 * pays (17..32 states: every primitive model; 33..64: up to 96 sequences per call), otherwise one wave walks one
 * sequence — with a SPARSE step (each lane gathers its own predecessors / successors) when no state of the model
 * has more than 8 of either, e.g. the multi-copy gene models; decided per model on the device, HMM_OPT_FORCE_DENSE = 1
 * forces the all-candidates step.  Zero entries of A are exact zeros in both steps; the two differ in rounding order only.
 * hmm_viterbi covers q <= hmm_viterbi_max_states() (64), hmm_loglik_grad q <= hmm_grad_max_states() (64). */
int hmm_max_states(void);
int hmm_scan_max_states(void);
int hmm_viterbi_max_states(void);
int hmm_grad_max_states(void);

/* Time-chunk length the engine will use for (k*b, L): a multiple of 16; 0 for the serial large-q path. */
int hmm_chunk_len(int k, int b, int L, int q);

/* Column width (64, 80 or 96) of the GEMM tile the serial large-q path uses for a batch of b
 * sequences per model and q > 64 states (the instantiation is chosen per shape so that the 256 CUs
 * get as even a share of tiles as possible); 0 when q is served by another path. */
int hmm_largeq_tile_cols(int b, int q);

size_t hmm_workspace_bytes(int op, int k, int b, int L, int q);

/*
 * Forward recursion.  Replaces _forward_recursion_impl (hmm_layer/MsaHMMLayer.py:227-282)
 * = BaseRNN time loop (hmm_layer/BaseRNN.py:217-227) over HmmCell.forward
 * (hmm_layer/MsaHmmCell.py:73-106, forward branch :102-103).
 *   log_alpha (k,b,L,q) or NULL : log alpha_t = log alpha_hat_t + sum_{s<=t} log c_s
 *   loglik    (k,b) fp64        : sum_t log c_t
 */
int hmm_forward(const float *A, const float *pi, const float *E,
                int k, int b, int L, int q, float eps,
                float *log_alpha, double *loglik,
                void *workspace, size_t workspace_bytes, void *stream);

/*
 * Backward recursion.  Replaces _backward_recursion_impl (hmm_layer/MsaHMMLayer.py:322-381)
 * = the reverse HmmCell (hmm_layer/MsaHmmCell.py:96-100) run over flipped time.
 *   log_beta (k,b,L,q), beta_{L-1} = 1.
 */
int hmm_backward(const float *A, const float *E,
                 int k, int b, int L, int q, float eps,
                 float *log_beta,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * State posteriors.  Replaces _state_posterior_log_probs_impl
 * (hmm_layer/MsaHMMLayer.py:422-521) = Bidirectional (hmm_layer/Bidirectional.py:113-164)
 * over the forward and reverse cells + chunk stitching via TotalProbabilityCell
 * (hmm_layer/TotalProbabilityCell.py:30-49, hmm_layer/MsaHMMLayer.py:285-319, 384-419).
 *   out    (k,b,L,q) : per `mode` (HMM_POST_*)
 *   loglik (k,b) fp64 or NULL
 * Posteriors are formed from the scaled per-position variables and renormalised per
 * position, which is algebraically the reference's log alpha + log beta - loglik
 * (hmm_layer/MsaHMMLayer.py:501-514) without its fp32 cancellation.
 *
 * The cell clamps the predicted state MIXTURE at eps every step (hmm_layer/MsaHmmCell.py:87-88); the
 * chunk operators of the scan are the exactly linear products of A diag(max(E, eps)) and know nothing of it.
 * For q <= 16 the engine therefore decides on the device, with no host round trip, which inputs the scan
 * may serve:
 *   - per model: the support of A (entries > eps) must be primitive (irreducible and aperiodic);
 *     reducible or periodic chains, states without incoming edges, all-zero rows (the reference's
 *     as-shipped matrices) and A = I go to serial kernels with the cell's exact step semantics;
 *   - per sequence: the posterior mass of CLAMP-BORN paths — paths through a component that a clamp of either
 *     cell lifted to eps — is exactly what separates the serial recursion from the clamp-free scan (posteriors
 *     and log-likelihood alike).  The in-chunk kernels, which do apply the clamps, sum it per chunk; sequences
 *     above 2e-6 (a tenth of the posteriors' stated tolerance) are recomputed serially — only WINDOWS of chunks
 *     around the ones that carry that mass, grown until the recursion has forgotten it (the cost of a flagged
 *     sequence is its flagged chunks plus the model's forgetting time, not its length), in every entry point:
 *     hmm_posterior; hmm_forward (forward cell's births, weighed with the chunk scan's backward vectors; log alpha
 *     after a window moves with the window's log-likelihood), hmm_backward (the mirror image, weighed with the
 *     forward vectors of a uniform start) and hmm_loglik_grad (forward cell's births).  Windows that grow into
 *     each other, and sequences with more than 16 of them, are redone whole.  Chunks whose operator columns went
 *     through the denormal range (two observations in a row that every path survives at the emission floor only)
 *     count as flagged.
 * hmm_exact_count() reports how many of the last call's sequences took the serial kernels.
 */
int hmm_posterior(const float *A, const float *pi, const float *E,
                  int k, int b, int L, int q, float eps, int mode,
                  float *out, double *loglik,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Diagnostics of the routing above: reads, from the workspace of a finished q <= 16 call (the
 * caller synchronises first), how many of its k*b sequences were served by the serial exact-clamp
 * kernels.  `op` and the shape are those of the call.  Returns the count or a negative error. */
long long hmm_exact_count(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes);
/* The same for a finished hmm_posterior call, q <= 16, in detail:
 *   detail[0] sequences that left the scan (= hmm_exact_count)
 *   detail[1] of those, sequences recomputed in windows (runs of chunks around the flagged ones)
 *   detail[2] the number of such windows
 *   detail[3] sequences recomputed whole because most of their chunks were flagged or because two of their
 *             windows grew into each other (sequences of models routed per model are in detail[0] only)
 *   detail[4] chunks (of hmm_chunk_len positions) the windows walked, their growth until the recursion had
 *             forgotten the clamp-born mass included */
int hmm_exact_detail(int k, int b, int L, int q, const void *workspace, size_t workspace_bytes, long long *detail);
/* The same for the last hmm_forward (HMM_OP_LOGLIK without log alpha, HMM_OP_FORWARD with) or hmm_backward
 * (HMM_OP_BACKWARD) call of this shape: the op selects the workspace layout.  HMM_OP_POSTERIOR = hmm_exact_detail. */
int hmm_exact_detail_op(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes,
                        long long *detail);
/* Test / diagnostic hook: the window table of sequence `seq` after the last call of `op` with this shape (q <= 16):
 * table[34] = count, spare, then (first chunk, chunks) pairs as the window kernels left them; shifts[24] (or null;
 * HMM_OP_FORWARD / HMM_OP_BACKWARD) = 16 per-window log-scale shifts as doubles, then 16 ints (last / first chunks);
 * psi[npsi] (or null) = the certificate's per-chunk sums.  Returns the number of chunks, or an error. */
int hmm_window_table(int op, int k, int b, int L, int q, const void *workspace, size_t workspace_bytes, int seq,
                     int *table, double *shifts, float *psi, int npsi);

/*
 * Viterbi state paths (max-plus scan).  The reference has none (only a docstring mention,
 * hmm_layer/MsaHmmCell.py:13, and the unused log_A_dense, :46-47); this entry point is what
 * a Viterbi over HmmCell's parameters needs: log A (k,q,q) (transitioner.make_log_A()),
 * log pi (k,q), log E (k,b,L,q) = log of cell.emission_probs() (clamped as the caller sees fit).
 * Scores are Q16 fixed point, Q(x) = rint(clip(x,-1024,1024)*65536), so the chunked scan is
 * bit-identical to the serial recursion; ties take the lowest state index (oracle/viterbi.py).
 *   path  (k,b,L) int32 : most probable state sequence
 *   score (k,b)   fp64  : its log-probability under the quantised model
 */
size_t hmm_viterbi_workspace_bytes(int k, int b, int L, int q);
int hmm_viterbi(const float *logA, const float *logpi, const float *logE,
                int k, int b, int L, int q,
                int32_t *path, double *score,
                void *workspace, size_t workspace_bytes, void *stream);

/*
 * Fused emission producer of the gene-prediction models.  Replaces GenePredHMMEmitter.forward
 * (hmm_layer/gene_pred_hmm_emitter.py:231-277, class part :93-121) and kmer.make_k_mers
 * (hmm_layer/kmer.py:3-47) for inference with one model:
 *   x           (b,L,s+5)  class probabilities followed by one-hot nucleotides A,C,G,T,N
 *   B           (rows,s)   softmax of the emission kernel (emitter.make_B())
 *   state_row   (q) int    kernel row feeding state j (intron parameter sharing, :115-116)
 *   codon       (2,nc,64)  left / right 3-mer tables (emitter.codon_probs, :198-217)
 *   state_codon (q) int    table row constraining state j, or -1 (free state: `free_value`, 1/4096)
 *   add                    added to the 3-mer factor (1e-7 when training, else 0, :260-261)
 *   n_mass                 1, or 2 to reproduce the reference's doubled N mass in right 3-mers
 *   E           (b,L,q)    emission probabilities, the engine's input
 */
int hmm_gene_emissions(const float *x, int b, int L, int s, const float *B, int rows,
                       const int *state_row, const float *codon, int nc, const int *state_codon, int q,
                       float free_value, float add, int n_mass, float *E, void *stream);

/*
 * Per-kernel timing for the roofline report (bench.py): the same computation as
 * hmm_posterior with every kernel launch bracketed by HIP events recorded on `stream`.
 * hmm_profile_read() waits for the recorded events, returns the summed milliseconds and
 * the launch count per kernel (arrays of HMM_KERNEL_COUNT) and resets the profile.
 */
#define HMM_KERNEL_REDUCE   0   /* chunk operators (MFMA matrix-product chain) */
#define HMM_KERNEL_SCAN     1   /* chunk-level prefix / suffix                 */
#define HMM_KERNEL_FORWARD  2   /* in-chunk forward pass, checkpoints          */
#define HMM_KERNEL_BACKWARD 3   /* in-chunk backward pass, posteriors          */
#define HMM_KERNEL_EXACT    4   /* routing + serial exact-clamp kernels (empty launches when nothing is routed) */
#define HMM_KERNEL_COUNT    5
void *hmm_profile_create(void);
void hmm_profile_destroy(void *profile);
int hmm_posterior_profiled(const float *A, const float *pi, const float *E,
                           int k, int b, int L, int q, float eps, int mode,
                           float *out, double *loglik,
                           void *workspace, size_t workspace_bytes, void *stream, void *profile);
int hmm_profile_read(void *profile, double *ms, long long *launches);

/*
 * Weighted log-likelihood aggregate.  Replaces MsaHmmLayer.apply_sequence_weights with
 * aggregate=True (hmm_layer/MsaHMMLayer.py:155-164): for each model the pair
 * (sum_b w*loglik, sum_b w) in fp64.  `weights` (k,b) fp32 or NULL (= ones).
 *   partial (k,2) fp64 device.  The cross-GPU step is one all-reduce(sum) of `partial`
 *   (done by the host wrapper over RCCL); mean over models follows on the host.
 */
int hmm_loglik_partials(const double *loglik, const float *weights, int k, int b,
                        double *partial, void *stream);

/*
 * The cross-GPU step of the same aggregate for hosts that do not have torch.distributed:
 * in-place all-reduce(sum) of `partial` (k,2) fp64 over `comm`, an ncclComm_t (RCCL) the HOST
 * created for its ranks, enqueued on `stream`.  The engine does not link RCCL: it calls the
 * ncclAllReduce of the RCCL library already loaded in the process (HMM_ERR_NO_RCCL if there is
 * none).  Afterwards every rank holds (sum over all ranks' sequences of w*loglik, sum of w) per
 * model; the weighted mean and the mean over models are the host's two divisions
 * (hmm_layer/MsaHMMLayer.py:160-164).
 */
int hmm_loglik_allreduce(void *comm, double *partial, int k, void *stream);

/*
 * Sequence-sharded posteriors: every rank owns one contiguous TIME slab of every sequence
 * (E_slab (k,b,Ls,q), Ls may differ between ranks) — for batches too small to be cut across GPUs.
 * The roles of TotalProbabilityCell.forward (hmm_layer/TotalProbabilityCell.py:30-49) and
 * _get_total_forward/backward_from_chunks (hmm_layer/MsaHMMLayer.py:285-319, 384-419) lifted across
 * devices.  q <= hmm_scan_max_states().  Protocol, R ranks, rank r (slabs in time order):
 *   1. hmm_seqshard_reduce(...)    -> slab_op (k,b,16,16) fp32, slab_exp (k,b,16) int32: the slab's
 *                                     operator per sequence (column n scaled by 2^-exp[n])
 *   2. host: all-gather slab_op / slab_exp over the ranks and lay them out (k,b,R,16,16) / (k,b,R,16)
 *   3. hmm_seqshard_posterior(...) -> out (k,b,Ls,q) per `mode`, loglik (k,b) = the WHOLE sequence's,
 *                                     phi_out (k,b) fp32 or NULL: this slab's share of the sequence's
 *                                     floor-transition bound (see hmm_posterior; +inf when A's support is
 *                                     not primitive).  The host sums phi over ranks; above 1e-6 the
 *                                     sequence needs the unsharded call (serial exact-clamp kernels).
 * seq_start: 1 on the rank that owns position 0 (r == 0), else 0.  The same workspace (same size query)
 * must be passed to steps 1 and 3: the chunk operators stay in it.
 */
size_t hmm_seqshard_workspace_bytes(int k, int b, int Ls, int q, int R);
int hmm_seqshard_reduce(const float *A, const float *E, int k, int b, int Ls, int q, float eps, int seq_start,
                        int R, float *slab_op, int *slab_exp,
                        void *workspace, size_t workspace_bytes, void *stream);
int hmm_seqshard_posterior(const float *A, const float *pi, const float *E, int k, int b, int Ls, int q, float eps,
                           int seq_start, const float *all_ops, const int *all_exps, int R, int r, int mode,
                           float *out, double *loglik, float *phi_out,
                           void *workspace, size_t workspace_bytes, void *stream);

/*
 * Gradient of the log-likelihoods (training).  The reference trains by autograd through the
 * Python time loop (hmm_layer/BaseRNN.py:217-227 over HmmCell.forward,
 * hmm_layer/MsaHmmCell.py:73-106); this entry point returns the same derivatives from one
 * forward-backward pass (Baum-Welch expectations), for q <= hmm_grad_max_states():
 *   grad_loglik (k,b) fp32 or NULL (= ones): d loss / d loglik[m][s], the upstream gradient
 *   dA  (k,q,q) : sum_s grad_loglik * d loglik / d A    (= sum_t xi_t(i,j) / A[i][j], dense)
 *   dpi (k,q)   : sum_s grad_loglik * d loglik / d pi
 *   dE  (k,b,L,q): grad_loglik * gamma / E; zero where the cell clamps E below eps
 *   loglik (k,b) fp64 or NULL
 * Entries of E / pi that the cell clamps to eps receive no gradient (torch.maximum semantics).
 * Sums over sequences and time run in a fixed order (fp32 per 16-chain tile, fp64 across tiles):
 * results are deterministic.
 */
size_t hmm_loglik_grad_workspace_bytes(int k, int b, int L, int q);
/* 17..64 states: how many sequences of the last call with this shape the whole-sequence sweeps served (all of
 * them, except for the compiled 29-state two-copy topology on up to 512 sequences, which is computed per chunk of
 * the 32-state scan plan under the routing of hmm_posterior); q <= 16: hmm_exact_count of that call. */
long long hmm_loglik_grad_serial_count(int k, int b, int L, int q, const void *workspace, size_t workspace_bytes);
int hmm_loglik_grad(const float *A, const float *pi, const float *E,
                    int k, int b, int L, int q, float eps, const float *grad_loglik,
                    float *dA, float *dpi, float *dE, double *loglik,
                    void *workspace, size_t workspace_bytes, void *stream);

/*
 * Gradient of a loss on the state posteriors (training through state_posterior_log_probs).  The
 * reference differentiates _state_posterior_log_probs_impl by autograd through its Python loops
 * (hmm_layer/MsaHMMLayer.py:422-521 called with training=True, tests/parallel_rnn_forward.py:70-80);
 * this is that reverse-mode computation as four sweeps per sequence (two value sweeps, two adjoint
 * sweeps; lane = state), for q <= hmm_posterior_grad_max_states() (64):
 *   mode      HMM_POST_PROB (out = gamma) or HMM_POST_LOG (out = log gamma)
 *   grad_out  (k,b,L,q) : d loss / d out
 *   dA (k,q,q), dpi (k,q), dE (k,b,L,q) : d loss / d A, pi, E; clamped entries receive nothing
 * Deterministic (fixed summation order).
 *
 * For q <= 16 and up to 4096 sequences the sweeps run per chunk of the scan plan, in parallel (the
 * adjoint recursions are affine in the adjoint vector: every chunk's map is measured, the maps are
 * scanned, the sweeps rerun from the true entering vectors; HMM_OPT_PGCHUNK).  Which sequences that
 * path may serve is decided on the device as for hmm_posterior — per model by the support of A, per
 * sequence by the floor-transition bound F = eps * sum_t 1 / <alpha_hat_t, R_t> <= 1e-6 — and in log
 * mode additionally by how much of the upstream gradient sits on states whose posterior is so small
 * that floor paths can matter for THEM: sum |G| min(1, F / gamma) <= 1e-4 sum |G|.  The remaining
 * sequences are redone by whole-sequence sweeps in the same call.  The compiled 29-state two-copy
 * topology is served the same way (rows of 32 lanes, up to 512 sequences); larger batches and other
 * models with q > 16 use the whole-sequence sweeps throughout.  One stated exception, as for
 * hmm_loglik_grad: dA entries of ABSENT edges (A = 0) weigh the adjoint of states that are improbable
 * where the edge would lead to them; across chunk boundaries that adjoint travels through chunk
 * operators whose eps floors are additive, and with emissions of ~1e-10 on the probable path such an
 * entry can be off by a percent (the whole-sequence sweeps, HMM_OPT_PGCHUNK = 0, have them to 1e-6).
 * The reference never reads them: its A is scattered from per-edge parameters.  hmm_posterior_grad_serial_count() reads, from the workspace of
 * a finished call with the same shape (the caller synchronises first), how many sequences those were.
 */
int hmm_posterior_grad_max_states(void);
size_t hmm_posterior_grad_workspace_bytes(int k, int b, int L, int q);
long long hmm_posterior_grad_serial_count(int k, int b, int L, int q, const void *workspace, size_t workspace_bytes);
int hmm_posterior_grad(const float *A, const float *pi, const float *E,
                       int k, int b, int L, int q, float eps, int mode, const float *grad_out,
                       float *dA, float *dpi, float *dE,
                       void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HMM_ENGINE_H */
