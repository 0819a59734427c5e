// extern "C" surface (include/sefd.h): plan life cycle, op execution, fused losses, Adam.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <map>
#include "../../include/sefd.h"
#include "dev_common.h"
#include "plan.h"
#include "tuning.h"

using namespace sefd;

struct sefd_plan {
  Plan* p;
  std::vector<std::string> names;
  // second stream + events for the off-critical-path lane (created on first use, owned by the plan)
  mutable hipStream_t side = nullptr, side2 = nullptr;   // side2: lane 3 (the layer-1 input GEMMs between the chunks of the forward recurrences)
  mutable hipStream_t mainq = nullptr;                    // CU-partition experiment (SEFD_MAIN_CUS): a library-owned, CU-masked stream that carries the main lane
  mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_side2 = nullptr, ev_main = nullptr;
  // host-mapped status word of THIS plan (created on first use: plans are also built on hosts without a GPU).  0 = fine; sticky once set
  // by a kernel that gave up (cluster LSTM hand-over timeout) until sefd_plan_status(clear = 1)
  mutable int* status = nullptr;
  mutable int* dstatus = nullptr;             // the same word in device memory (what the guarded Adam reads)
  // knob GRAPH=1 (round 6 experiment, profiles/r06_tuning_notes.md section 12): whole-phase runs replayed as an instantiated hipGraph captured from the
  // executor's own multi-stream launch sequence; keyed by (phase, flags, stream, arena bases); the first two runs of a key execute normally (lazy
  // allocations and function attributes happen there, not inside a capture)
  struct GraphKey { int phase, flags; void* stream; void* arena[A_COUNT]; bool operator<(const GraphKey& o) const { return std::memcmp(this, &o, sizeof(*this)) < 0; } };
  struct GraphEntry { int calls = 0; hipGraphExec_t exec = nullptr; bool failed = false; };
  mutable std::map<GraphKey, GraphEntry> graphs;
};
static int* plan_status_word(const sefd_plan* h) {
  if (!h->status) {
    int* q = nullptr;
    int* dq = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&q), sizeof(int), hipHostMallocMapped) != hipSuccess) return nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dq), 256) != hipSuccess || hipMemset(dq, 0, 256) != hipSuccess) { (void)hipHostFree(q); return nullptr; }
    *q = 0;
    h->status = q;
    h->dstatus = dq;
  }
  return h->status;
}

static_assert(sizeof(sefd_model_config) == sizeof(ModelConfig), "config mirror out of sync");

extern "C" {

sefd_plan* sefd_plan_create(const sefd_model_config* cfg) {
  ModelConfig mc;
  std::memcpy(&mc, cfg, sizeof(mc));
  sefd_plan* h = new sefd_plan();
  h->p = build_plan(mc);
  for (auto& kv : h->p->bufs) h->names.push_back(kv.first);
  return h;
}
void sefd_plan_destroy(sefd_plan* h) {
  if (!h) return;
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  if (h->side2) { (void)hipStreamSynchronize(h->side2); (void)hipStreamDestroy(h->side2); }
  if (h->mainq) { (void)hipStreamSynchronize(h->mainq); (void)hipStreamDestroy(h->mainq); }
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->ev_side2) (void)hipEventDestroy(h->ev_side2);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  if (h->status) (void)hipHostFree(h->status);
  if (h->dstatus) (void)hipFree(h->dstatus);
  delete h->p;
  delete h;
}
const char* sefd_plan_error(const sefd_plan* h) { return h->p->error.c_str(); }
int64_t sefd_plan_arena_bytes(const sefd_plan* h, int a) { return (a >= 0 && a < A_COUNT) ? h->p->arena_bytes[a] : -1; }
int32_t sefd_plan_frames(const sefd_plan* h) { return h->p->T; }
static const std::vector<ParamInfo>& pv(const sefd_plan* h, int kind) { return kind == 0 ? h->p->params : h->p->state; }
int32_t sefd_plan_num_params(const sefd_plan* h, int kind) { return (int32_t)pv(h, kind).size(); }
const char* sefd_plan_param_name(const sefd_plan* h, int kind, int i) { return pv(h, kind)[i].name.c_str(); }
int64_t sefd_plan_param_offset(const sefd_plan* h, int kind, int i) { return pv(h, kind)[i].off; }
int64_t sefd_plan_param_numel(const sefd_plan* h, int kind, int i) { return pv(h, kind)[i].numel; }
int32_t sefd_plan_param_shape(const sefd_plan* h, int kind, int i, int64_t* s4) {
  const auto& sh = pv(h, kind)[i].shape;
  for (size_t k = 0; k < sh.size() && k < 4; ++k) s4[k] = sh[k];
  return (int32_t)sh.size();
}
int32_t sefd_plan_buffer(const sefd_plan* h, const char* name, int32_t* arena, int64_t* off, int64_t* bytes, int32_t* dtype) {
  auto it = h->p->bufs.find(name);
  if (it == h->p->bufs.end()) return -1;
  *arena = std::strncmp(name, "io.", 3) == 0 ? A_IO : A_WS;
  *off = it->second.off; *bytes = it->second.bytes; *dtype = it->second.dtype;
  return 0;
}
int32_t sefd_plan_num_buffers(const sefd_plan* h) { return (int32_t)h->names.size(); }
const char* sefd_plan_buffer_name(const sefd_plan* h, int i) { return h->names[i].c_str(); }
const void* sefd_plan_const_data(const sefd_plan* h) { return h->p->consts.data(); }
int32_t sefd_plan_num_syncs(const sefd_plan* h) { return (int32_t)h->p->syncs.size(); }
int32_t sefd_plan_sync(const sefd_plan* h, int i, int32_t* phase, int32_t* op, int32_t* arena, int64_t* off, int64_t* count, int32_t* dtype) {
  if (i < 0 || i >= (int)h->p->syncs.size()) return -1;
  const SyncPoint& sp = h->p->syncs[i];
  *phase = sp.phase; *op = sp.op; *arena = sp.buf.arena; *off = sp.buf.off; *count = sp.count; *dtype = sp.dtype;
  return 0;
}
int32_t sefd_plan_num_ops(const sefd_plan* h, int phase) { return (int32_t)(phase == 0 ? h->p->fwd.size() : h->p->bwd.size()); }
const void* sefd_plan_ops(const sefd_plan* h, int phase) { return phase == 0 ? h->p->fwd.data() : h->p->bwd.data(); }
int32_t sefd_op_size(void) { return (int32_t)sizeof(Op); }
int32_t sefd_plan_op_info(const sefd_plan* h, int phase, int i, int64_t* o) {
  const std::vector<Op>& ops = phase == 0 ? h->p->fwd : h->p->bwd;
  if (i < 0 || i >= (int)ops.size()) return -1;
  const Op& op = ops[i];
  for (int k = 0; k < 8; ++k) o[k] = 0;
  o[0] = op.kind; o[1] = op.tag;
  if (op.kind == OP_RUNGEMM || op.kind == OP_WGRAD) {
    const RunGemm& g = op.g;
    int64_t K = 0;
    for (int s = 0; s < g.nseg; ++s) K += g.seg[s].len;
    o[2] = g.M; o[3] = g.N; o[4] = K; o[5] = g.xdt | ((int64_t)g.flags << 8);        // low byte: operand dtype; above: RunGemm flags
    o[6] = 2 * (int64_t)g.M * g.N * K;                                     // algorithmic flops (true K, true N)
    // algorithmic bytes: every source element once is not well defined for overlapping runs; report A-row + y + w traffic
    o[7] = (int64_t)g.M * g.N * esize(g.ydt) + (int64_t)g.N * K * esize(g.xdt);
    if (g.flags & kRunEnc0) o[7] += (int64_t)(g.M / g.Fo) * g.tstride[0] * 4;     // first encoder layer on the spectrum: + every frame of it, once (HBM-bound kernels: enc0.hip)
  } else if (op.kind == OP_LSTM_FWD || op.kind == OP_LSTM_BWD) {            // recurrent gate GEMM: rows x 4H x H per launch
    const LstmRec& r = op.lstm;
    const int64_t steps = op.kind == OP_LSTM_FWD && r.t1 > 0 ? r.t1 - r.t0 : r.T;
    const int64_t rows = (int64_t)r.G * r.B * steps;
    o[2] = rows; o[3] = 4 * r.H; o[4] = r.H; o[5] = r.hdt;
    o[6] = 2 * rows * 4 * r.H * r.H;
    o[7] = rows * (4 * r.H * 4 + 4 * r.H * 4 + r.H * (4 + esize(r.hdt)));   // gx read, gates written, c and h written
  } else if (op.kind == OP_STFT_FFT) {                                     // HBM-bound: samples read + half spectra written
    const StftFft& f = op.fft;
    o[2] = (int64_t)f.B * f.T; o[3] = 257; o[4] = 512;
    o[7] = (int64_t)f.B * f.L * 4 + (int64_t)f.B * f.T * 258 * 8;
    if (f.lp.arena >= 0) o[7] += (int64_t)f.B * f.T * 258 * 8 * esize(f.lp_dt);     // + the channel-padded copy for the first encoder layer (8 channels per bin)
  } else if (op.kind == OP_ISTFT_FFT) {
    o[2] = op.ifft.nframes; o[3] = op.ifft.W; o[4] = 512;
    o[7] = op.ifft.nframes * (258 * 8 + (int64_t)op.ifft.W * 4);
  }
  return 0;
}

const int32_t* sefd_plan_status_word(const sefd_plan* h) { return (h && plan_status_word(h)) ? reinterpret_cast<const int32_t*>(h->dstatus) : nullptr; }
int32_t sefd_plan_status(const sefd_plan* h, int32_t clear) {
  if (!h || !h->status) return 0;
  if (!clear) return __atomic_load_n(h->status, __ATOMIC_RELAXED);
  (void)hipDeviceSynchronize();                // no kernel of the plan is still about to set it
  (void)hipMemset(h->dstatus, 0, sizeof(int));
  return __atomic_exchange_n(h->status, 0, __ATOMIC_RELAXED);
}
int32_t sefd_plan_status_set(const sefd_plan* h) {   // what a kernel that gives up does, from the host (tests)
  if (!h || !plan_status_word(h)) return -1;
  const int one = 1;
  if (hipMemcpy(h->dstatus, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -2;
  __atomic_store_n(h->status, 1, __ATOMIC_RELAXED);
  return 0;
}
int32_t sefd_plan_grad_bucket(const sefd_plan* h, int32_t* op, int64_t* elem) {
  if (!h || h->p->bucket_op < 0) return -1;
  *op = h->p->bucket_op; *elem = h->p->bucket_elem;
  return 0;
}
int32_t sefd_plan_grad_bucket_range(const sefd_plan* h, int32_t* op, int64_t* lo, int64_t* hi) {
  if (!h || h->p->bucket_op < 0) return -1;
  *op = h->p->bucket_op; *lo = h->p->bucket_elem;
  *hi = h->p->bucket_end >= 0 ? h->p->bucket_end : h->p->arena_bytes[A_GRAD] / 4;
  return 0;
}

static int32_t plan_run(const sefd_plan* h, int phase, int first, int last, void* const* arenas, void* stream, int at, void (*cb)(void*), void* ctx,
                        std::vector<hipEvent_t>* tev = nullptr, int flags = 0);

static int32_t plan_run_graph(const sefd_plan* h, int phase, void* const* arenas, void* stream, int flags);
static bool graph_knob() { const char* e = tune_str("GRAPH"); return e && atoi(e) == 1; }      // read per call: the equivalence test flips it
int32_t sefd_plan_run(const sefd_plan* h, int phase, int first, int last, void* const* arenas, void* stream) {
  if (h && first <= 0 && last < 0 && graph_knob() && !tune_str("NO_OVERLAP")) return plan_run_graph(h, phase, arenas, stream, 0);
  return plan_run(h, phase, first, last, arenas, stream, -1, nullptr, nullptr);
}
int32_t sefd_plan_run_cb(const sefd_plan* h, int phase, void* const* arenas, void* stream, int at, void (*cb)(void*), void* ctx) {
  return plan_run(h, phase, 0, -1, arenas, stream, at, cb, ctx);
}
static int32_t plan_run_graph(const sefd_plan* h, int phase, void* const* arenas, void* stream, int flags) {
  sefd_plan::GraphKey key;
  std::memset(&key, 0, sizeof(key));
  key.phase = phase; key.flags = flags; key.stream = stream;
  for (int a = 0; a < A_COUNT; ++a) key.arena[a] = arenas[a];
  sefd_plan::GraphEntry& e = h->graphs[key];
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (e.exec) {
    int* status = plan_status_word(h);
    if (status && __atomic_load_n(status, __ATOMIC_RELAXED) != 0) return -5;
    return hipGraphLaunch(e.exec, st) == hipSuccess ? 0 : -2;
  }
  if (e.failed || ++e.calls < 3) return plan_run(h, phase, 0, -1, arenas, stream, -1, nullptr, nullptr, nullptr, flags);
  hipGraph_t g = nullptr;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { e.failed = true; (void)hipGetLastError(); return plan_run(h, phase, 0, -1, arenas, stream, -1, nullptr, nullptr, nullptr, flags); }
  const int32_t rc = plan_run(h, phase, 0, -1, arenas, stream, -1, nullptr, nullptr, nullptr, flags);
  const hipError_t ec = hipStreamEndCapture(st, &g);
  if (rc != 0 || ec != hipSuccess || !g || hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0) != hipSuccess) {
    e.failed = true; e.exec = nullptr;
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    fprintf(stderr, "sefd: GRAPH=1: capture of phase %d failed (rc %d, %s): running the launch sequence\n", phase, (int)rc, hipGetErrorString(ec));
    return plan_run(h, phase, 0, -1, arenas, stream, -1, nullptr, nullptr, nullptr, flags);
  }
  (void)hipGraphDestroy(g);
  return hipGraphLaunch(e.exec, st) == hipSuccess ? 0 : -2;
}
int32_t sefd_plan_run_flags(const sefd_plan* h, int phase, void* const* arenas, void* stream, int flags, int at, void (*cb)(void*), void* ctx) {
  if (graph_knob() && h && !cb && !tune_str("NO_OVERLAP")) return plan_run_graph(h, phase, arenas, stream, flags);
  return plan_run(h, phase, 0, -1, arenas, stream, at, cb, ctx, nullptr, flags);
}
// Measurement: the whole phase in its REAL two-lane schedule, with a HIP event recorded before and after every op on the stream the op
// is launched on; synchronises, then ms[i] = duration of op i in situ (next to whatever the other lane runs).  n = number of ops.
int32_t sefd_plan_run_timed(const sefd_plan* h, int phase, void* const* arenas, void* stream, float* ms, int32_t n) {
  const int nops = sefd_plan_num_ops(h, phase);
  if (n < nops) return -1;
  std::vector<hipEvent_t> ev(2 * (size_t)nops, nullptr);
  for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return -3;
  const int32_t rc = plan_run(h, phase, 0, -1, arenas, stream, -1, nullptr, nullptr, &ev);
  (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
  if (h->side) (void)hipStreamSynchronize(h->side);
  for (int i = 0; i < nops; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]) != hipSuccess) t = -1.f;
    ms[i] = t;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

}  // extern "C"

static int32_t plan_run(const sefd_plan* h, int phase, int first, int last, void* const* arenas, void* stream, int at, void (*cb)(void*), void* ctx,
                        std::vector<hipEvent_t>* tev, int flags) {
  if (!h || !h->p->error.empty()) return -1;
  int* status = plan_status_word(h);
  if (status && __atomic_load_n(status, __ATOMIC_RELAXED) != 0) return -5;   // an earlier launch of THIS plan gave up waiting (cluster LSTM): sticky until cleared
  const std::vector<Op>& ops = phase == 0 ? h->p->fwd : h->p->bwd;
  if (first < 0) first = 0;
  if (last < 0 || last > (int)ops.size()) last = (int)ops.size();
  ArenaBases ab;
  for (int a = 0; a < A_COUNT; ++a) ab.p[a] = reinterpret_cast<char*>(arenas[a]);
  ab.status = status;
  ab.dstatus = h->dstatus;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // SEFD_RUN_WAVE_ONLY: the caller's loss reads the waveform only (the fused train step without a perceptual term): the [B, NF, T] copies of
  // the enhanced spectrum and the accumulation of their (all-zero) gradients are not launched
  const bool wave_only = (flags & SEFD_RUN_WAVE_ONLY) != 0;
  auto launch = [&](const Op& op, hipStream_t s) {
    if (wave_only && (op.kind == OP_SPECOUT_FWD || op.kind == OP_SPECOUT_BWD) && op.so.mode == 0 && (op.kind == OP_SPECOUT_FWD || op.so.accumulate)) return;
    const size_t idx = (size_t)(&op - ops.data());
    if (tev) (void)hipEventRecord((*tev)[2 * idx], s);
    if (op.kind == OP_RUNGEMM) launch_rungemm(op.g, ab, s);
    else if (op.kind == OP_WGRAD) launch_wgrad(op.g, ab, s);
    else launch_misc(op, ab, s);
    if (tev) (void)hipEventRecord((*tev)[2 * idx + 1], s);
  };
  // Two stacked row-block LSTM layers (FullSubNet's sub-band model) that follow each other on the main stream become ONE launch
  // (lstm_rows.hip lstm_fwd_rows_pair_kernel): true = ops i and i + 1 were both issued
  auto launch_pair = [&](int i, hipStream_t s) -> bool {
    if (i + 1 >= last || i == at) return false;
    const Op &a = ops[i], &b = ops[i + 1];
    if (a.kind != OP_LSTM_FWD || b.kind != OP_LSTM_FWD || a.lane != 0 || b.lane != 0 || b.join != 0) return false;
    if (a.lstm.impl != 1 || b.lstm.impl != 1 || a.lstm.hdt != DT_BF16) return false;
    if (tev) (void)hipEventRecord((*tev)[2 * i], s);
    if (!launch_lstm_rows_pair(a.lstm, b.lstm, ab, s)) return false;
    if (tev) { (void)hipEventRecord((*tev)[2 * i + 1], s); (void)hipEventRecord((*tev)[2 * i + 2], s); (void)hipEventRecord((*tev)[2 * i + 3], s); }
    return true;
  };
  // Two-lane execution of a whole phase: lane-1 ops (weight gradients, the folds of their row-split partial sums, the early UNPACK) in front
  // of the first LSTM backward are held back and issued on the side stream right after that kernel, so they fill the CUs the recurrence
  // leaves idle; later lane-1 ops follow at their program position, except the ones marked kOpHold, which wait for the NEXT recurrence
  // launch (FullSubNet).  The main stream waits for the lane at the ops that carry `join` and at every main-stream UNPACK.
  // Partial runs (tests, per-op timing) and SEFD_NO_OVERLAP=1 execute everything in program order on `stream`.
  const bool no_overlap = tune_str("NO_OVERLAP") != nullptr;      // read per call: the schedule-equivalence test flips it between two steps
  bool two_lane = !no_overlap && first == 0 && last == (int)ops.size();
  if (two_lane) {
    bool any1 = false, any2 = false, lstm = false;
    for (const Op& op : ops) { any1 |= op.lane == 1; any2 |= op.lane >= 2; lstm |= op.kind == OP_LSTM_BWD; }
    two_lane = any2 || (any1 && lstm);
  }
  if (!two_lane) {                                       // program order on one stream is always a valid schedule
    for (int i = first; i < last; ++i) {
      if (launch_pair(i, st)) { ++i; if (i == at && cb) cb(ctx); continue; }
      launch(ops[i], st);
      if (i == at && cb) cb(ctx);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (!h->side) {
    // (a lowest-priority side stream was measured: 14.44 / 14.57 vs 14.33 / 14.42 ms per step with equal priorities - kept equal)
    // SEFD_SIDE_CUS=N (tuning): the second lane's stream is confined to N of the chip's CUs, spread evenly over the XCDs
    // (hipExtStreamCreateWithCUMask), so that the weight gradients stop time-slicing every CU with the main lane's 160 KB-LDS GEMMs.  Measured
    // (profiles/r05_tuning_notes.md): no N beats the unmasked stream - the default stays unmasked.
    // Round 6: BOTH lanes on library-created masked queues.  SEFD_MAIN_CUS=M puts the main lane of a whole-phase run on a second
    // masked stream with the M CUs the side lane's mask leaves out first (M + N > 256: the two masks overlap on the remaining CUs), forked from / joined
    // to the caller's stream by events.  Measured: profiles/r06_tuning_notes.md.
    const int side_cus = tune_str("SIDE_CUS") ? atoi(tune_str("SIDE_CUS")) : 0;
    const int main_cus = tune_str("MAIN_CUS") ? atoi(tune_str("MAIN_CUS")) : 0;
    uint32_t smask[8] = {0};
    if (side_cus > 0 && side_cus < 256) {
      for (int i = 0; i < 256; ++i)
        if ((i + 1) * side_cus / 256 > i * side_cus / 256) smask[i >> 5] |= 1u << (i & 31);
      if (hipExtStreamCreateWithCUMask(&h->side, 8, smask) != hipSuccess) return -3;
    } else if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) return -3;
    if (main_cus > 0 && main_cus <= 256) {
      uint32_t mmask[8] = {0};
      int n = 0;
      for (int i = 0; i < 256 && n < main_cus; ++i) if (!(smask[i >> 5] >> (i & 31) & 1)) { mmask[i >> 5] |= 1u << (i & 31); ++n; }   // the side lane's complement first
      for (int i = 0; i < 256 && n < main_cus; ++i) if (!(mmask[i >> 5] >> (i & 31) & 1)) { mmask[i >> 5] |= 1u << (i & 31); ++n; }
      if (hipExtStreamCreateWithCUMask(&h->mainq, 8, mmask) != hipSuccess) return -3;
      if (hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming) != hipSuccess) return -3;
    }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return -3;
    if (hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) return -3;
    if (hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking) != hipSuccess) return -3;
    if (hipEventCreateWithFlags(&h->ev_side2, hipEventDisableTiming) != hipSuccess) return -3;
  }
  hipStream_t caller = st;
  const bool use_mainq = h->mainq && phase == 1;          // the backward phase only: the forward's lanes are its LSTM chunks, the main lane needs every CU there
  if (use_mainq) {                                        // the main lane moves to the masked queue: ordered behind the caller's stream here, joined back below
    (void)hipEventRecord(h->ev_main, caller);
    (void)hipStreamWaitEvent(h->mainq, h->ev_main, 0);
    st = h->mainq;
  }
  auto to_caller = [&]() {                                // everything the main lane has been given so far is ordered in front of the caller's next work
    if (!use_mainq) return;
    (void)hipEventRecord(h->ev_main, st);
    (void)hipStreamWaitEvent(caller, h->ev_main, 0);
  };
  std::vector<int> held;
  bool has_lstm_bwd = false;
  for (const Op& op : ops) has_lstm_bwd |= op.kind == OP_LSTM_BWD;
  int last_lstm = -1;                                    // lane-1 ops behind the last recurrence are never held back
  for (int i = 0; i < (int)ops.size(); ++i) if (ops[i].kind == OP_LSTM_BWD) last_lstm = i;
  bool forked = !has_lstm_bwd;                           // no LSTM backward in this phase: lane-1 ops are never held back
  bool side_busy = false;                                // the side stream holds work the main stream has not waited for
  bool side2_busy = false;                               // the third stream holds work the side stream has not waited for
  bool fork_fresh = false;                               // ev_fork was recorded and the main stream has been given nothing since
  auto fork = [&]() { if (!fork_fresh) { (void)hipEventRecord(h->ev_fork, st); fork_fresh = true; } };
  auto side_launch = [&](const Op& op) {                 // program order up to here is satisfied on the main stream
    fork();
    (void)hipStreamWaitEvent(h->side, h->ev_fork, 0);
    if (side2_busy) {                                    // ... and on the third stream (lane-2 ops read what lane-3 ops in front of them wrote)
      (void)hipEventRecord(h->ev_side2, h->side2);
      (void)hipStreamWaitEvent(h->side, h->ev_side2, 0);
      side2_busy = false;
    }
    launch(op, h->side);
    side_busy = true;
  };
  // lane 3: ordered behind the main stream's work so far and behind earlier lane-3 ops only - NOT behind the side stream, so the planner
  // gives it ops that read main-stream results alone (layer-1 input GEMM of chunk k beside layer 1's recurrence over chunk k - 1)
  auto side2_launch = [&](const Op& op) {
    fork();
    (void)hipStreamWaitEvent(h->side2, h->ev_fork, 0);
    launch(op, h->side2);
    side2_busy = true;
  };
  auto join = [&]() {
    if (side2_busy) {
      (void)hipEventRecord(h->ev_side2, h->side2);
      (void)hipStreamWaitEvent(st, h->ev_side2, 0);
      side2_busy = false;
    }
    if (!side_busy) return;
    (void)hipEventRecord(h->ev_join, h->side);
    (void)hipStreamWaitEvent(st, h->ev_join, 0);
    side_busy = false;
  };
  // SEFD_HOLD_SKIP=K (tuning): the first K lane-1 ops in front of the first recurrence are NOT held back but issued at their program position
  // (beside the decoder's BatchNorm / dgrad chain); the rest still waits for the recurrence.  Measured: profiles/r05_tuning_notes.md.
  static const int hold_skip = tune_str("HOLD_SKIP") ? atoi(tune_str("HOLD_SKIP")) : 0;
  // SEFD_NO_OPHOLD=1 (tuning): lane-1 ops marked kOpHold behind the first recurrence are issued at their program position instead of behind the next recurrence
  static const bool no_ophold = tune_str("NO_OPHOLD") && atoi(tune_str("NO_OPHOLD")) == 1;
  int lane1_seen = 0;
  for (int i = first; i < last; ++i) {
    const Op& op = ops[i];
    // held: every lane-1 op in front of the first recurrence, and the ones marked kOpHold, wait for the next recurrence launch
    if (op.lane == 1 && i < last_lstm && (!forked || (op.join == kOpHold && !no_ophold))) {
      if (!forked && lane1_seen++ < hold_skip) { side_launch(op); continue; }
      held.push_back(i); continue;
    }
    if (op.lane == 3) { side2_launch(op); continue; }
    if (op.lane == 1 || op.lane == 2) { side_launch(op); continue; }
    if (op.kind == OP_LSTM_BWD && (!forked || !held.empty())) {
      fork_fresh = false;
      fork();                                            // everything the held ops read has been produced before this point
      launch(op, st);                                    // the recurrence takes its CUs first
      fork_fresh = false;
      (void)hipStreamWaitEvent(h->side, h->ev_fork, 0);
      for (int j : held) launch(ops[j], h->side);
      side_busy = side_busy || !held.empty();
      held.clear();
      forked = true;
      continue;
    }
    if (op.join == 1 || (op.kind == OP_UNPACK && op.join != kOpNoJoin)) join();   // UNPACK gathers gradient partials: needs the side lane's results
                                                         // (kOpNoJoin: a bucket whose partials all come from the main stream - FullSubNet's full-band model)
    if (launch_pair(i, st)) { ++i; fork_fresh = false; if (i == at && cb) { to_caller(); cb(ctx); } continue; }
    launch(op, st);
    fork_fresh = false;
    if (i == at && cb) { to_caller(); cb(ctx); }         // e.g. the first gradient bucket is complete: the caller starts its all-reduce
  }
  join();
  to_caller();
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


// =============================================================================================== losses
// Per-utterance inner products in one pass over est/tgt (HBM-bound: 2 x 4 x L bytes per utterance), then a
// one-workgroup finalize that turns them into the scalar loss AND the two coefficients (ca, cb) of the analytic
// gradient  d loss / d est[b][n] = ca[b] * est[b][n] + cb[b] * tgt[b][n]  (every loss of tools_for_loss.py:17-94 has this form).
namespace {
constexpr int kLossBlk = 16;     // workgroups per utterance

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* est, const float* tgt, int L, float* part) {
  const int b = blockIdx.y, blk = blockIdx.x;
  const float4* e4 = reinterpret_cast<const float4*>(est + (int64_t)b * L);
  const float4* t4 = reinterpret_cast<const float4*>(tgt + (int64_t)b * L);
  const int n4 = (L % 4 == 0) ? L / 4 : 0;          // rows are 16-byte aligned only when L is a multiple of 4 (spectra: L = T = 483)
  float see = 0.f, set = 0.f, stt = 0.f;
  for (int i = blk * 256 + threadIdx.x; i < n4; i += kLossBlk * 256) {
    const float4 e = e4[i], t = t4[i];
    see += e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
    set += e.x * t.x + e.y * t.y + e.z * t.z + e.w * t.w;
    stt += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
  }
  if (blk == 0 || n4 == 0)
    for (int i = n4 * 4 + (n4 == 0 ? blk * 256 : 0) + threadIdx.x; i < L; i += (n4 == 0 ? kLossBlk * 256 : 256)) {
      const float e = est[(int64_t)b * L + i], t = tgt[(int64_t)b * L + i];
      see += e * e; set += e * t; stt += t * t;
    }
  __shared__ float r[3][4];
  see = wave_sum(see); set = wave_sum(set); stt = wave_sum(stt);
  if ((threadIdx.x & 63) == 0) { r[0][threadIdx.x >> 6] = see; r[1][threadIdx.x >> 6] = set; r[2][threadIdx.x >> 6] = stt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((int64_t)b * kLossBlk + blk) * 3;
    o[0] = r[0][0] + r[0][1] + r[0][2] + r[0][3];
    o[1] = r[1][0] + r[1][1] + r[1][2] + r[1][3];
    o[2] = r[2][0] + r[2][1] + r[2][2] + r[2][3];
  }
}

// ws layout: part [B][kLossBlk][3] | coef [B][2] | terms[B]
__global__ void loss_finalize_kernel(int kind, int B, int L, float* ws, float* loss_out) {
  float* part = ws;
  float* coef = ws + (int64_t)B * kLossBlk * 3;
  float* term = coef + 2 * B;
  const float eps = 1e-8f;
  const float k10 = 4.342944819032518f;          // 10 / ln(10)
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float see = 0.f, set = 0.f, stt = 0.f;
    for (int k = 0; k < kLossBlk; ++k) { see += part[(b * kLossBlk + k) * 3]; set += part[(b * kLossBlk + k) * 3 + 1]; stt += part[(b * kLossBlk + k) * 3 + 2]; }
    float v = 0.f, ca = 0.f, cb = 0.f;
    if (kind == SEFD_LOSS_MSE) {                 // F.mse_loss(est, tgt)
      v = (see - 2.f * set + stt) / ((float)B * (float)L);
      ca = 2.f / ((float)B * (float)L); cb = -ca;
    } else if (kind == SEFD_LOSS_SDR) {          // -mean 10 log10(stt^2 / (D^2 + eps)), D = |t - e|^2   (tools_for_loss.py:29-33)
      const float Dn = see - 2.f * set + stt;
      v = -k10 * logf(stt * stt / (Dn * Dn + eps)) / B;
      const float kk = k10 / B * 4.f * Dn / (Dn * Dn + eps);
      ca = kk; cb = -kk;
    } else if (kind == SEFD_LOSS_SISNR) {        // -mean 10 log10(|a t|^2 / (|e - a t|^2 + eps) + eps), a = <e,t>/(<t,t>+eps)  (:36-44)
      const float den = stt + eps;
      const float a = set / den;
      const float Tn = a * a * stt;
      const float Nn = see - 2.f * a * set + a * a * stt;
      const float Rr = Tn / (Nn + eps) + eps;
      v = -k10 * logf(Rr) / B;
      const float A1 = 1.f / (Nn + eps), A2 = Tn / ((Nn + eps) * (Nn + eps));
      const float g = -k10 / (B * Rr);
      ca = g * (-2.f * A2);
      cb = g * (A1 * 2.f * a * stt / den - A2 * (-2.f * a + 2.f * (a * stt - set) / den));
    } else {                                     // SI-SDR: ratio_b = P/N + eps; loss = -10 log10(mean_b ratio_b + eps)   (:47-94)
      const float a = set / stt + eps;
      const float Pn = a * a * stt;
      const float Nn = see - 2.f * a * set + a * a * stt;
      v = Pn / Nn + eps;                         // ratio_b ; the log is applied after the batch mean
      // d ratio / d e = (2 a t) / N - P/N^2 * (2(e - a t) + 2 t (a stt - set)/stt)
      ca = -Pn / (Nn * Nn) * 2.f;
      cb = 2.f * a / Nn - Pn / (Nn * Nn) * (-2.f * a + 2.f * (a * stt - set) / stt);
    }
    term[b] = v; coef[2 * b] = ca; coef[2 * b + 1] = cb;
    acc += v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float total = red[0];
  if (kind == SEFD_LOSS_SISDR) {
    const float m = total / B;
    const float g = -k10 / (m + eps) / B;
    for (int b = threadIdx.x; b < B; b += blockDim.x) { coef[2 * b] *= g; coef[2 * b + 1] *= g; }
    if (threadIdx.x == 0) {
      loss_out[0] = -k10 * logf(m + eps);
      // data parallel (sefd_loss_dp_finish): this rank's (sum of ratios, rows), once to be all-reduced in place and once to stay
      float* dp = term + B;
      dp[0] = total; dp[1] = (float)B; dp[2] = total; dp[3] = (float)B;
    }
  } else if (threadIdx.x == 0) {
    loss_out[0] = total;
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* est, const float* tgt, int B, int L, const float* ws,
                                                       const float* gscale, float* g) {
  const float* coef = ws + (int64_t)B * kLossBlk * 3;
  const float gs = gscale ? gscale[0] : 1.f;
  const int64_t n = (int64_t)B * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L);
    g[i] = gs * (coef[2 * b] * est[i] + coef[2 * b + 1] * tgt[i]);
  }
}

// ---- short rows (FullSubNet.loss, models.py:674-682: the reductions of tools_for_loss.py run over the LAST axis of [B, F, T, 2], i.e. over
// millions of two-element rows, and trainer.py:107 hands the network output over in the `target` slot).  One thread per row; every term is a
// function f(see, set, stt) of the three inner products, so  d f / d est = 2 f_see est + f_set tgt,  d f / d tgt = f_set est + 2 f_stt tgt -
// the gradient with respect to EITHER argument.  The differences (t - e, e - a t) are formed element-wise as the reference forms them, not
// expanded (two-element rows of near-equal vectors cancel badly when expanded).
constexpr int kRowsMaxL = 16;
constexpr int kRowsMaxBlk = 4096;
struct RowTerms { float v, fsee, fset, fstt; };

__device__ __forceinline__ RowTerms row_terms(int kind, const float* e, const float* t, int L, float invR, float invRL) {
  const float eps = 1e-8f, k10 = 4.342944819032518f;
  float see = 0.f, set = 0.f, stt = 0.f, dd = 0.f;
  for (int i = 0; i < L; ++i) { see += e[i] * e[i]; set += e[i] * t[i]; stt += t[i] * t[i]; const float d = t[i] - e[i]; dd += d * d; }
  RowTerms r;
  if (kind == SEFD_LOSS_MSE) {
    r.v = dd * invRL; r.fsee = invRL; r.fset = -2.f * invRL; r.fstt = invRL;
  } else if (kind == SEFD_LOSS_SDR) {            // -10 log10(stt^2 / (D^2 + eps)) / R, D = |t - e|^2, s1 = t (tools_for_loss.py:29-33)
    r.v = -k10 * logf(stt * stt / (dd * dd + eps)) * invR;
    const float q = k10 * invR * 2.f * dd / (dd * dd + eps);
    r.fsee = q; r.fset = -2.f * q; r.fstt = -k10 * invR * 2.f / stt + q;
  } else {
    const bool snr = kind == SEFD_LOSS_SISNR;
    const float den = snr ? stt + eps : stt;
    const float a = snr ? set / den : set / den + eps;   // si_snr: <e,t>/(<t,t>+eps) ; si_sdr: <t,e>/<t,t> + eps
    const float Tn = a * a * stt;
    float Nn = 0.f;
    for (int i = 0; i < L; ++i) { const float d = e[i] - a * t[i]; Nn += d * d; }
    const float dNa = -2.f * set + 2.f * a * stt;          // d Nn / d a
    const float da_dstt = snr ? -a / den : -set / (stt * stt);
    if (snr) {                                             // -10 log10(Tn / (Nn + eps) + eps) / R    (:36-44)
      const float Rr = Tn / (Nn + eps) + eps;
      r.v = -k10 * logf(Rr) * invR;
      const float g = -k10 * invR / Rr, A1 = 1.f / (Nn + eps), A2 = Tn / ((Nn + eps) * (Nn + eps));
      r.fsee = g * (-A2);
      r.fset = g * (A1 * 2.f * a * stt / den - A2 * (-2.f * a + dNa / den));
      r.fstt = g * (A1 * (a * a + 2.f * a * stt * da_dstt) - A2 * (a * a + dNa * da_dstt));
    } else {                                               // ratio = P / N + eps; the log follows the mean over rows   (:47-94)
      r.v = Tn / Nn + eps;
      const float q = Tn / (Nn * Nn);
      r.fsee = -q;
      r.fset = 2.f * a / Nn - q * (-2.f * a + dNa / den);
      r.fstt = (a * a + 2.f * a * stt * da_dstt) / Nn - q * (a * a + dNa * da_dstt);
    }
  }
  return r;
}

__global__ __launch_bounds__(256) void loss_rows_reduce_kernel(int kind, const float* est, const float* tgt, int64_t R, int L, float* part) {
  const float invR = 1.f / (float)R, invRL = invR / (float)L;
  float acc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256)
    acc += row_terms(kind, est + r * L, tgt + r * L, L, invR, invRL).v;
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ws: part [nblk] | G (SI-SDR: the factor -k10 / (mean + eps) / R of every row's ratio gradient; 1 otherwise)
__global__ void loss_rows_finalize_kernel(int kind, int64_t R, int nblk, float* ws, float* loss_out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += ws[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float eps = 1e-8f, k10 = 4.342944819032518f;
    if (kind == SEFD_LOSS_SISDR) {
      const float m = red[0] / (float)R;
      loss_out[0] = -k10 * logf(m + eps);
      ws[nblk] = -k10 / (m + eps) / (float)R;
      ws[nblk + 1] = red[0]; ws[nblk + 2] = (float)R; ws[nblk + 3] = red[0]; ws[nblk + 4] = (float)R;     // see sefd_loss_dp_finish
    } else {
      loss_out[0] = red[0];
      ws[nblk] = 1.f;
    }
  }
}

__global__ __launch_bounds__(256) void loss_rows_grad_kernel(int kind, const float* est, const float* tgt, int64_t R, int L, const float* ws, int nblk,
                                                            const float* gscale, float* gest, float* gtgt) {
  const float invR = 1.f / (float)R, invRL = invR / (float)L;
  const float gs = (gscale ? gscale[0] : 1.f) * ws[nblk];
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R; r += (int64_t)gridDim.x * 256) {
    const float* e = est + r * L;
    const float* t = tgt + r * L;
    const RowTerms q = row_terms(kind, e, t, L, invR, invRL);
    for (int i = 0; i < L; ++i) {
      if (gest) gest[r * L + i] = gs * (2.f * q.fsee * e[i] + q.fset * t[i]);
      if (gtgt) gtgt[r * L + i] = gs * (q.fset * e[i] + 2.f * q.fstt * t[i]);
    }
  }
}
// SI-SDR over data-parallel ranks: tools_for_loss.py:91-94 takes the mean of the ratios over the WHOLE batch before the log, so the
// per-rank loss is not a term of a sum.  dp = { global sum of ratios, global rows (all-reduced in place by the host), this rank's sum, this rank's
// rows }.  Loss <- -10 log10(global mean + eps); the row coefficients, already scaled with the rank's own mean by the finalize kernel, are
// re-scaled to (d loss / d ratio) * world - the exchange sums the ranks' gradients and Adam multiplies by 1 / world.
__global__ void loss_dp_finish_kernel(float* coef, int64_t ncoef, const float* dp, float world, float* loss_out) {
  const float eps = 1e-8f, k10 = 4.342944819032518f;
  const float m = dp[0] / dp[1], ml = dp[2] / dp[3];
  const float f = ((ml + eps) * dp[3] * world) / ((m + eps) * dp[1]);
  for (int64_t i = threadIdx.x; i < ncoef; i += blockDim.x) coef[i] *= f;
  if (threadIdx.x == 0 && loss_out) loss_out[0] = -k10 * logf(m + eps);
}
__global__ void status_poison_kernel(const int32_t* dstatus, float* elem) {
  if (__hip_atomic_load(dstatus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) *elem = __builtin_nanf("");
}
static int rows_blocks(int64_t R) { const int64_t n = (R + 255) / 256; return (int)(n < kRowsMaxBlk ? n : kRowsMaxBlk); }

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float bc2_sqrt,
                                                  float b1, float b2, float eps, float gscale, const int32_t* skip, const float* skip_nan) {
  // skip_nan (data parallel): an element of the all-reduced gradient that a rank whose plan status was set replaced by NaN before the
  // exchange (status_poison_kernel) - the sum carries it to every rank, so ALL replicas skip the same step and stay identical
  if (skip_nan) { const float q = *skip_nan; if (q != q) return; }
  // skip: the device copy of the status word of the plan that produced g.  Set = a kernel of this step gave up and g is garbage:
  // parameters and moments stay as they are (the host raises at its next look at the host-mapped copy)
  if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
}  // namespace

extern "C" {
int64_t sefd_loss_ws_floats(int32_t B) { return (int64_t)B * kLossBlk * 3 + 3 * (int64_t)B + 16; }

int32_t sefd_loss_forward(int kind, const float* est, const float* tgt, int32_t B, int32_t L, float* ws, float* loss_out, void* stream) {
  if (kind < 0 || kind > 3 || B < 1 || L < 1) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(kLossBlk, B), dim3(256), 0, st, est, tgt, L, ws);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, kind, B, L, ws, loss_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_loss_backward(int kind, const float* est, const float* tgt, int32_t B, int32_t L, const float* ws,
                           const float* grad_scale, float* grad_est, void* stream) {
  (void)kind;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)B * L;
  int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(grid), dim3(256), 0, st, est, tgt, B, L, ws, grad_scale, grad_est);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int64_t sefd_loss_rows_ws_floats(int64_t R) { return rows_blocks(R) + 16; }
int32_t sefd_loss_rows_forward(int kind, const float* est, const float* tgt, int64_t R, int32_t L, float* ws, float* loss_out, void* stream) {
  if (kind < 0 || kind > 3 || R < 1 || L < 1 || L > kRowsMaxL) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = rows_blocks(R);
  hipLaunchKernelGGL(loss_rows_reduce_kernel, dim3(nblk), dim3(256), 0, st, kind, est, tgt, R, L, ws);
  hipLaunchKernelGGL(loss_rows_finalize_kernel, dim3(1), dim3(256), 0, st, kind, R, nblk, ws, loss_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_loss_rows_backward(int kind, const float* est, const float* tgt, int64_t R, int32_t L, const float* ws, const float* grad_scale,
                                float* grad_est, float* grad_tgt, void* stream) {
  if (kind < 0 || kind > 3 || R < 1 || L < 1 || L > kRowsMaxL) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nblk = rows_blocks(R);
  hipLaunchKernelGGL(loss_rows_grad_kernel, dim3(nblk), dim3(256), 0, st, kind, est, tgt, R, L, ws, nblk, grad_scale, grad_est, grad_tgt);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int64_t sefd_loss_dp_offset(int64_t n, int32_t rows) { return rows ? (int64_t)rows_blocks(n) + 1 : n * kLossBlk * 3 + 3 * n; }
int32_t sefd_loss_dp_finish(int32_t rows, int64_t n, float* ws, int32_t world, float* loss_out, void* stream) {
  if (n < 1 || world < 1) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* dp = ws + sefd_loss_dp_offset(n, rows);
  if (rows) hipLaunchKernelGGL(loss_dp_finish_kernel, dim3(1), dim3(64), 0, st, ws + rows_blocks(n), (int64_t)1, dp, (float)world, loss_out);
  else hipLaunchKernelGGL(loss_dp_finish_kernel, dim3(1), dim3(256), 0, st, ws + n * kLossBlk * 3, 2 * n, dp, (float)world, loss_out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                       float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return sefd_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, grad_scale, nullptr, stream);
}
int32_t sefd_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                               float lr, float beta1, float beta2, float eps, float grad_scale, const int32_t* skip_if_set, void* stream) {
  return sefd_adam_step_guarded_dp(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, grad_scale, skip_if_set, nullptr, stream);
}
int32_t sefd_plan_status_poison(const sefd_plan* h, float* grad_elem, void* stream) {
  if (!h || !grad_elem || !plan_status_word(h)) return -1;
  hipLaunchKernelGGL(status_poison_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), h->dstatus, grad_elem);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
int32_t sefd_adam_step_guarded_dp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step, float lr, float beta1,
                                  float beta2, float eps, float grad_scale, const int32_t* skip_if_set, const float* skip_if_nan, void* stream) {
  if (n < 1 || step < 1) return -1;
  const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, (float)(lr / bc1), (float)std::sqrt(bc2),
                     beta1, beta2, eps, grad_scale, skip_if_set, skip_if_nan);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}void sefd_tuning_set(const char* knob, const char* value) { if (knob) sefd::tune_set(knob, value); }
const char* sefd_tuning_get(const char* knob) { return knob ? sefd::tune_str(knob) : nullptr; }
void sefd_tuning_clear(void) { sefd::tune_clear(); }


