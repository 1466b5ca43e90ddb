// Internal header of libacrmi.so: the context, the error / device-guard helpers and what the translation units of the C
// ABI share.  acrmi.hip = context life cycle, options, the fused entry points; acrmi_program.hip = program validation,
// dependency schedule and replay (one stream / parallel lanes); acrmi_ops.hip = the stand-alone operators;
// acrmi_comm.hip = RCCL.
#pragma once
#include "../../include/acrmi.h"
#include "kernels.h"

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace acrmi;

// ---- op dependencies and stream lanes ----------------------------------------------------------------
constexpr int MAX_LANES = 8;
// ACRMI_OPT_LANES = 0: measured on MI355X (tools/lanes_check.py) - batch 1: 6.97 ms on one stream, 4.63 on four;
// batch 32: 23.2 / 22.5 / 22.2 ms on one / two / four; batch 64: 46.9 ms on one, 45.6 on two (the second lane fills the
// tails and pipeline fills of the first), 45.8 on three
constexpr int AUTO_LANES_SMALL = 4, AUTO_LANES_LARGE = 2, AUTO_SMALL_BATCH = 32;
struct Schedule {
  std::vector<int> order;               // active op indices, program order (a topological order)
  std::vector<std::vector<int>> deps;   // per op: earlier ops it must wait for (RAW / WAR / WAW on buffer ids)
  std::vector<char> leaf;               // per op: no later op depends on it
  // multi-stream form (ACRMI_OPT_LANES): lane per op, events of other lanes' ops to wait for, event to record
  int n_lanes = 0;
  std::vector<int> lane;
  std::vector<std::vector<int>> wait;
  std::vector<char> signal;
};
struct acrmi_ctx {
  int device = 0;
  std::string err;
  float* weights = nullptr;     // the packed blob on the device - owned, or another context's (acrmi_share_weights)
  size_t n_weights = 0;
  std::atomic<int>* weights_ref = nullptr;   // host-side use count of `weights` shared by the contexts that hold it (null: no blob).  Atomic so that contexts may be DESTROYED from different threads; acrmi_share_weights(c, donor) against a concurrent destroy / reload of the donor is NOT safe (the count could reach 0 between its read and its increment): callers serialize those (engine.EnginePool builds its contexts on one thread)
  std::vector<acrmi_buffer_desc> bufs;
  std::vector<float*> buf_ptr;
  std::vector<acrmi_op> ops;
  acrmi_head_layout heads{};
  bool have_program = false;
  int max_batch = 0;
  float* att_ws = nullptr;      // attention-pool workspace
  size_t att_ws_floats = 0;
  int* picks = nullptr;         // point heads: decoded centers per frame [max_batch,4]
  int* gate_buf = nullptr;      // ACRMI_OPT_BATCH_PRIOR: the batch-wide prior decision per frame [max_batch] (acrmi_prior_gate)
  bool batch_prior = false;     // ACRMI_OPT_BATCH_PRIOR: acrmi_forward applies the reference's batch-wide prior rules at B > 1
  unsigned* range_flag = nullptr;   // 'fp16x3' programs (algo 6): set by conv_x3 / conv_x3p when an activation left the f16 range
                                    // (acrmi_check_range reads and clears it; acrmi_decode poisons the slots while it is set)
  bool point_heads = false;     // ACRMI_OPT_POINT_HEADS
  Schedule sched[2][2];         // [point][0: small batches / 1: large batches]
  // ACRMI_OPT_LANES: independent chains of the program on parallel HIP streams (lane 0 = the caller's stream)
  int want_lanes = 0;           // 0 = by batch size (AUTO_LANES_*)
  // ACRMI_OPT_LANE_PLAN: lane per op from MEASURED op times (acrmi_profile_ops at a small batch stores them here), small-
  // batch schedules only; empty = the structural heuristic
  bool lane_plan = false;       // (explicit: Engine.tune_lanes / the host switches it on after profiling)
  std::vector<float> op_ms[2];  // [point]: per-op milliseconds of the last small-batch profile
  hipStream_t lanes[MAX_LANES] = {};
  hipEvent_t fork_ev = nullptr, join_ev[MAX_LANES] = {};
  std::vector<hipEvent_t> op_ev;
  // split-K convolutions (ACRMI_CONV_SPLITK): partial tiles + arrival counters, one set per lane (ops of one lane are
  // stream-ordered; ops of different lanes may overlap)
  float* split_ws[MAX_LANES] = {};
  unsigned* split_cnt[MAX_LANES] = {};
  size_t split_ws_floats = 0, split_counters = 0;
  ManoTables mano[2]{};
  bool have_mano[2] = {false, false};
  float* mano_allocs[2][11] = {};  // 6 fp32 tables + 3 f16 copies + 2 f16 residual (lo) tables per side
  bool mano_f16 = false;           // ACRMI_OPT_MANO_FP16
  // options the reference reads from its config (acr/config.py): centermap_conf_thresh (acr/result_parser.py:241),
  // align_idx / mano_mesh_root_align (acr/mano_wrapper.py:19-33), -t temporal_optimization + smooth_coeff (acr/main.py:45-47)
  float conf_thresh = 0.35f;
  int center_idx = 9;           // < 0: no root alignment
  bool temporal = false;
  float smooth_coeff = 4.0f;
  float* smooth_state = nullptr;   // [2][3][64] floats + 2 ints (One-Euro state of one video stream)
  // multi-GPU: RCCL communicator created by acrmi_comm_init
  void* comm = nullptr;
  int comm_ranks = 0;
};

// Every entry point runs on the context's device and leaves the caller's current device as it found it (a process
// may hold contexts on several GPUs).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

// last error of calls without a context (acrmi_create, the stand-alone operators)
extern std::string g_err;
int fail(acrmi_ctx* c, int code, const char* fmt, ...);
#define ON_DEVICE(c)                                                                                   \
  DeviceGuard guard_((c)->device);                                                                     \
  if (guard_.err != hipSuccess) return fail(c, ACRMI_EHIP, "hipSetDevice(%d): %s", (c)->device, hipGetErrorString(guard_.err))
#define HIPCHK(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(c, ACRMI_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)


// ---- shared between the translation units ------------------------------------------------------------------------
void free_program(acrmi_ctx* c);                       // acrmi_program.hip
void release_weights(acrmi_ctx* c);                    // acrmi.hip: drops this context's hold on its weight blob
void comm_destroy(acrmi_ctx* c);                       // acrmi_comm.hip
int run_op(acrmi_ctx* c, const acrmi_op& op, const uint8_t* img, int B, hipStream_t s, int lane = 0);
bool op_active(const acrmi_op& op, bool point);
void build_schedule(acrmi_ctx* c, bool point, bool large);
int run_program(acrmi_ctx* c, const uint8_t* img, int B, void* stream, bool point, int first_op = 0);
int decode_maps_impl(const float* l_center, const float* r_center, int center_cs, const float* l_params, const float* r_params,
                     int params_cs, const float* l_prior, const float* r_prior, int prior_cs, int B, float conf_thresh,
                     const int32_t* prior_gate, const unsigned* poison, float* slots, void* stream);      // acrmi_ops.hip
