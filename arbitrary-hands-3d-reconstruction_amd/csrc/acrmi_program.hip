// libacrmi.so: the op program - validation (acrmi_set_program), dependency schedule and lanes, replay.
#include "acrmi_ctx.h"

void free_program(acrmi_ctx* c) {
  for (hipEvent_t e : c->op_ev)
    if (e) (void)hipEventDestroy(e);
  c->op_ev.clear();
  for (float* p : c->buf_ptr)
    if (p) (void)hipFree(p);
  c->buf_ptr.clear();
  c->bufs.clear();
  c->ops.clear();
  for (int l = 0; l < MAX_LANES; ++l) {
    if (c->split_ws[l]) (void)hipFree(c->split_ws[l]);
    if (c->split_cnt[l]) (void)hipFree(c->split_cnt[l]);
    c->split_ws[l] = nullptr; c->split_cnt[l] = nullptr;
  }
  c->split_ws_floats = c->split_counters = 0;
  if (c->att_ws) (void)hipFree(c->att_ws);
  c->att_ws = nullptr;
  if (c->picks) (void)hipFree(c->picks);
  c->picks = nullptr;
  if (c->gate_buf) (void)hipFree(c->gate_buf);
  c->gate_buf = nullptr;
  if (c->range_flag) (void)hipFree(c->range_flag);
  c->range_flag = nullptr;
  c->have_program = false;
}


int run_op(acrmi_ctx* c, const acrmi_op& op, const uint8_t* img, int B, hipStream_t s, int lane) {
  auto ptr = [&](int id) -> float* { return id >= 0 ? c->buf_ptr[id] : nullptr; };
  auto desc = [&](int id) -> const acrmi_buffer_desc& { return c->bufs[id]; };
  // buffer `id` advanced by `coff` ELEMENTS of its storage type (the pointers stay typed float*: they are opaque here)
  auto eptr = [&](int id, int coff) -> float* {
    if (id < 0) return nullptr;
    return reinterpret_cast<float*>(reinterpret_cast<char*>(c->buf_ptr[id]) + (size_t)coff * (c->bufs[id].dtype ? 2 : 4));
  };
  switch (op.kind) {
    case ACRMI_OP_U8NORM: {
      const auto& d = desc(op.out_buf);
      HIPCHK(c, launch_u8norm(img, (long)B * d.h * d.w, ptr(op.out_buf), s));
      return ACRMI_OK;
    }
    case ACRMI_OP_STEM: {
      const auto& d = desc(op.out_buf);
      if (op.ksize == 7) {     // ResNet stem (stem7.hip)
        if (d.dtype)
          HIPCHK(c, launch_stem7_h16(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf),
                                     d.cs, op.out_coff, op.relu, d.dtype, s));
        else
          HIPCHK(c, launch_stem7(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf), d.cs,
                                 op.out_coff, op.relu, s));
        return ACRMI_OK;
      }
      if (d.dtype)
        HIPCHK(c, launch_stem_h16(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf),
                                  d.cs, op.out_coff, op.relu, d.dtype, s));
      else
        HIPCHK(c, launch_stem(img, B, 2 * d.h, 2 * d.w, c->weights + op.w_off, c->weights + op.b_off, ptr(op.out_buf), d.cs,
                              op.out_coff, op.relu, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_CONV: {
      const auto& di = desc(op.in_buf);
      const auto& dout = desc(op.out_buf);
      ConvArgs a{};
      a.in = ptr(op.in_buf);
      a.w = c->weights + op.w_off;
      a.bias = op.bias_per_frame ? ptr(op.aux_buf) : c->weights + op.b_off;
      a.res = ptr(op.res_buf);
      a.out = ptr(op.out_buf);
      a.B = B; a.H = di.h; a.W = di.w; a.Ho = dout.h; a.Wo = dout.w;
      a.in_cs = di.cs; a.in_coff = op.in_coff; a.Cin = op.cin;
      a.out_cs = dout.cs; a.out_coff = op.out_coff; a.Cout = op.cout;
      a.res_cs = op.res_buf >= 0 ? desc(op.res_buf).cs : 0; a.res_coff = op.res_coff;
      if (op.flags & ACRMI_CONV_SPLITK) {     // groups = K-slices of one convolution; workspace of the lane this op runs on
        a.splitk = 1; a.split_ws = c->split_ws[lane]; a.split_cnt = c->split_cnt[lane];
      }
      if (op.flags & ACRMI_CONV_BIAS_MAP) {   // position-bias map in the weight blob, added to every frame
        a.res = c->weights + op.w_off2; a.res_cs = (op.groups * op.cout + 3) / 4 * 4; a.res_coff = 0; a.res_bcast = 1;
      }
      a.ks = op.ksize; a.stride = op.stride; a.relu = op.relu; a.groups = op.groups;
      a.cin8 = (op.cin + 7) / 8;
      a.n_tiles = op.cout <= 32 ? 1 : ((op.cout + 63) / 64) * 2;
      a.bias_fstride = op.bias_per_frame ? desc(op.aux_buf).cs : 0;
      a.nxt = op.nterms;      // extra residual terms of a stride-2 convolution (the HR fuse folded into its epilogue)
      for (int t = 0; t < op.nterms && t < 3; ++t) {
        a.xt[t] = ptr(op.term_buf[t]);
        a.xt_cs[t] = desc(op.term_buf[t]).cs; a.xt_coff[t] = op.term_coff[t]; a.xt_shift[t] = op.term_shift[t];
      }
      if (op.flags & ACRMI_CONV_DUAL) {      // the full-resolution HR fuse sum as a second output (conv_wino3's store waves)
        a.out2 = ptr(op.aux_buf); a.out2_cs = desc(op.aux_buf).cs; a.out2_coff = 0;
      }
      a.algo = op.flags & 7;
      a.range_flag = a.algo == 6 ? c->range_flag : nullptr;      // f16 operand halves: |x| must stay inside the f16 range
      a.dtype = di.dtype;                                    // 16-bit input: conv_h16.hip
      a.out_f32 = di.dtype != ACRMI_DT_F32 && dout.dtype == ACRMI_DT_F32;
      HIPCHK(c, launch_conv(a, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_FUSESUM: {
      const auto& dout = desc(op.out_buf);
      FuseArgs f{};
      f.nterms = op.nterms; f.B = B; f.H = dout.h; f.W = dout.w; f.C = op.cout; f.out_cs = dout.cs; f.relu = op.relu;
      f.out = eptr(op.out_buf, op.out_coff);
      for (int t = 0; t < op.nterms; ++t) {
        f.term[t] = eptr(op.term_buf[t], op.term_coff[t]);
        f.cs[t] = desc(op.term_buf[t]).cs;
        f.shift[t] = op.term_shift[t];
      }
      if (dout.dtype) HIPCHK(c, launch_fuse_sum_h16(f, dout.dtype, s));
      else HIPCHK(c, launch_fuse_sum(f, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_BILINEAR2X: {
      const auto& di = desc(op.in_buf);
      if (di.dtype)
        HIPCHK(c, launch_bilinear2x_h16(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                        desc(op.out_buf).cs, op.out_coff, di.dtype, s));
      else
        HIPCHK(c, launch_bilinear2x(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                    desc(op.out_buf).cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_PAIR1X1: {
      const auto& di = desc(op.in_buf);
      HIPCHK(c, launch_pair1x1(ptr(op.in_buf), di.cs, op.in_coff, ptr(op.res_buf), desc(op.res_buf).cs, op.res_coff, ptr(op.out_buf),
                               desc(op.out_buf).cs, op.out_coff, ptr(op.aux_buf), desc(op.aux_buf).cs, 0, c->weights + op.w_off,
                               (long)B * di.h * di.w, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_MAXPOOL: {
      const auto& di = desc(op.in_buf);
      if (di.dtype)
        HIPCHK(c, launch_maxpool3s2_h16(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                        desc(op.out_buf).cs, op.out_coff, di.dtype, s));
      else
        HIPCHK(c, launch_maxpool3s2(ptr(op.in_buf), B, di.h, di.w, di.cs, op.in_coff, op.cin, ptr(op.out_buf),
                                    desc(op.out_buf).cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_POW11: {
      const auto& d = desc(op.out_buf);
      if (d.dtype) HIPCHK(c, launch_pow11_h16(ptr(op.out_buf), (long)B * d.h * d.w, d.cs, op.out_coff, d.dtype, s));
      else HIPCHK(c, launch_pow11(ptr(op.out_buf), (long)B * d.h * d.w, d.cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_ATTPOOL: {
      const auto& ds = desc(op.in_buf);     // segm logits
      const auto& df = desc(op.res_buf);    // features
      HIPCHK(c, launch_attpool(ptr(op.in_buf), ds.cs, eptr(op.res_buf, op.res_coff), df.cs, op.cin, B, df.h, df.w,
                               c->att_ws, ptr(op.out_buf), s, df.dtype));
      return ACRMI_OK;
    }
    case ACRMI_OP_PAREBIAS: {
      PareArgs p{};
      p.pooled = ptr(op.in_buf);
      p.lc_w = c->weights + op.w_off;
      p.lin_w = c->weights + op.w_off2;
      p.lin_b = c->weights + op.b_off2;
      p.mix_wp = c->weights + op.w_off3;
      p.mix_b = c->weights + op.b_off;
      p.out = ptr(op.out_buf);
      p.B = B; p.C = op.cin; p.part0 = op.flags; p.out_stride = desc(op.out_buf).cs;
      HIPCHK(c, launch_parebias(p, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_COORDFILL: {
      const auto& d = desc(op.out_buf);
      if (d.dtype) HIPCHK(c, launch_coordfill_h16(ptr(op.out_buf), c->max_batch, d.h, d.w, d.cs, op.out_coff, d.dtype, s));
      else HIPCHK(c, launch_coordfill(ptr(op.out_buf), c->max_batch, d.h, d.w, d.cs, op.out_coff, s));
      return ACRMI_OK;
    }
    case ACRMI_OP_POINTHEADS: {
      const acrmi_head_layout& h = c->heads;
      const int side = op.flags & 1;
      PointArgs p{};
      p.x34 = ptr(op.in_buf); p.x_cs = desc(op.in_buf).cs;
      p.center[0] = ptr(h.center_buf[0]); p.center[1] = ptr(h.center_buf[1]); p.center_cs = desc(h.center_buf[0]).cs;
      p.w = c->weights + op.w_off;
      p.mix_w = c->weights + op.w_off2;
      p.bias = ptr(op.aux_buf); p.bias_stride = desc(op.aux_buf).cs;
      p.p109 = ptr(op.res_buf); p.p109_cs = desc(op.res_buf).cs;
      p.prior = ptr(h.prior_buf[side]); p.prior_cs = desc(h.prior_buf[side]).cs;
      p.final_ = ptr(op.out_buf); p.final_cs = desc(op.out_buf).cs;
      p.picks = c->picks; p.side = side; p.B = B; p.thresh = c->conf_thresh;
      p.prior_when_both = c->batch_prior ? 1 : 0;
      HIPCHK(c, launch_point_heads(p, s));
      return ACRMI_OK;
    }
    default:
      return fail(c, ACRMI_EINVAL, "unknown op kind %d", op.kind);
  }
}

// ops of the other head variant are skipped
bool op_active(const acrmi_op& op, bool point) {
  return op.kind != ACRMI_OP_COORDFILL && op.mode != (point ? ACRMI_MODE_DENSE : ACRMI_MODE_POINT);
}

// Buffers an op reads / writes (whole buffers: channel slices of one buffer are ordered conservatively).
// Pseudo-buffers n_bufs / n_bufs+1 stand for the attention-pool and center-pick workspaces.
static void op_rw(const acrmi_ctx* c, const acrmi_op& op, std::vector<int>& R, std::vector<int>& W) {
  R.clear(); W.clear();
  const int n_bufs = (int)c->bufs.size();
  auto r = [&](int id) { if (id >= 0) R.push_back(id); };
  auto w = [&](int id) { if (id >= 0) W.push_back(id); };
  switch (op.kind) {
    case ACRMI_OP_U8NORM: case ACRMI_OP_STEM: w(op.out_buf); break;
    case ACRMI_OP_CONV:
      r(op.in_buf); r(op.res_buf); if (op.bias_per_frame) r(op.aux_buf);
      for (int t = 0; t < op.nterms; ++t) r(op.term_buf[t]);
      w(op.out_buf);
      if (op.flags & ACRMI_CONV_DUAL) w(op.aux_buf);
      break;
    case ACRMI_OP_FUSESUM: for (int t = 0; t < op.nterms; ++t) r(op.term_buf[t]); w(op.out_buf); break;
    case ACRMI_OP_BILINEAR2X: case ACRMI_OP_MAXPOOL: r(op.in_buf); w(op.out_buf); break;
    case ACRMI_OP_PAIR1X1: r(op.in_buf); r(op.res_buf); w(op.out_buf); w(op.aux_buf); break;
    case ACRMI_OP_POW11: r(op.out_buf); w(op.out_buf); break;
    case ACRMI_OP_ATTPOOL: r(op.in_buf); r(op.res_buf); w(op.out_buf); w(n_bufs); break;
    case ACRMI_OP_PAREBIAS: r(op.in_buf); w(op.out_buf); break;
    case ACRMI_OP_POINTHEADS:
      r(op.in_buf); r(op.aux_buf); r(c->heads.center_buf[0]); r(c->heads.center_buf[1]);
      w(op.res_buf); w(op.out_buf); w(c->heads.prior_buf[op.flags & 1]); w(n_bufs + 1);
      break;
    default: break;
  }
}

// Dependencies between ops: RAW/WAR/WAW hazards on buffer ids (ids are reused for disjoint lifetimes, which
// the WAR edges respect).  The program order is a topological order.
void build_schedule(acrmi_ctx* c, bool point, bool large) {
  Schedule& S = c->sched[point ? 1 : 0][large ? 1 : 0];
  const int n = (int)c->ops.size(), nb = (int)c->bufs.size() + 2;
  S = Schedule();
  S.deps.assign(n, std::vector<int>());
  S.leaf.assign(n, 1);
  std::vector<int> last_writer(nb, -1), R, W;
  std::vector<std::vector<int>> readers(nb);
  for (int j = 0; j < n; ++j) {
    if (!op_active(c->ops[j], point)) continue;
    S.order.push_back(j);
    op_rw(c, c->ops[j], R, W);
    std::vector<int>& deps = S.deps[j];
    auto dep = [&](int i) { if (i >= 0 && i != j && std::find(deps.begin(), deps.end(), i) == deps.end()) deps.push_back(i); };
    for (int b : R) dep(last_writer[b]);
    for (int b : W) { dep(last_writer[b]); for (int i : readers[b]) dep(i); }
    for (int b : R) readers[b].push_back(j);
    for (int b : W) { last_writer[b] = j; readers[b].clear(); }
    for (int d : deps) S.leaf[d] = 0;
  }
  const int want = c->want_lanes > 0 ? c->want_lanes : (large ? AUTO_LANES_LARGE : AUTO_LANES_SMALL);
  const int max_lanes = std::max(1, std::min(want, MAX_LANES));
  S.lane.assign(n, 0);
  S.wait.assign(n, std::vector<int>());
  S.signal.assign(n, 0);
  std::vector<int> lane_tail(max_lanes, -1);
  const std::vector<float>& ms = c->op_ms[point ? 1 : 0];
  const bool planned = c->lane_plan && !large && max_lanes > 1 && (int)ms.size() == n;
  // Planned form (small batches, where a launch leaves most CUs idle and concurrent lanes really overlap): list
  // scheduling with the measured op times - every op, in program order, goes to the lane where it can START first.
  // Cost model measured on MI355X (tools/cross_stream_wait.py, tools/critical_path.py): a dependent kernel in the same
  // stream starts ~4.5 us after its producer ends, through an event on another stream ~21 us after (WAIT_MS below is the
  // difference); every other lane waited for costs the consumer's queue one barrier packet (SYNC_MS).
  constexpr float SYNC_MS = 0.002f, EVENT_MS = 0.002f;   // (EVENT_MS: what a profiled time includes)
  static const float WAIT_MS = experiment_env("ACRMI_PLAN_WAIT_US") ? 1e-3f * (float)atof(experiment_env("ACRMI_PLAN_WAIT_US")) : 0.016f;
  std::vector<float> fin(planned ? n : 0, 0.f), lane_free(max_lanes, 0.f);
  static const bool rank_order = !(experiment_env("ACRMI_PLAN_RANK") && atoi(experiment_env("ACRMI_PLAN_RANK")) == 0);
  if (planned && rank_order) {
    // Round 6: the ops are PLANNED AND ENQUEUED in order of their upward rank - the measured time of the
    // longest chain from the op to the end of the program - instead of program order.  Program order walks the four HRNet
    // branches round-robin, so the greedy earliest-start rule hands whichever lane is free to whatever op comes next and the
    // critical chain (101 launches of the two low-resolution branches + the head towers) keeps changing lanes, each change a
    // ~16 us cross-stream wait on the path.  In rank order the critical chain claims its lane first and stays on it; the
    // side chains fill the other lanes and their cross-lane edges have slack.  Any order that respects S.deps is a valid
    // enqueue order (every pair of ops that touch one buffer is ordered by an edge).  tools/sched_sim.py (the same cost
    // model, offline): batch 1 on four lanes 2.20 -> 2.05 ms simulated.
    std::vector<float> ru(n, 0.f);
    std::vector<std::vector<int>> succ(n);
    for (int j : S.order)
      for (int d : S.deps[j]) succ[d].push_back(j);
    for (int k = (int)S.order.size() - 1; k >= 0; --k) {
      const int j = S.order[k];
      float m = 0.f;
      for (int s2 : succ[j]) m = std::max(m, ru[s2]);
      ru[j] = m + std::max(ms[j], 1e-4f);      // (strictly larger than every successor's: the order stays topological)
    }
    std::stable_sort(S.order.begin(), S.order.end(), [&](int x, int y) { return ru[x] > ru[y]; });
  }
  std::vector<int> pos(n, -1);               // position in the enqueue order (a lane runs its ops in this order)
  for (int k = 0; k < (int)S.order.size(); ++k) pos[S.order[k]] = k;
  for (int j : S.order) {
    int lane = -1;
    if (planned) {
      float best = 0.f;
      bool best_prod = false;
      for (int l = 0; l < max_lanes; ++l) {
        float start = lane_free[l];
        unsigned others = 0;
        bool prod = false;
        for (int d : S.deps[j]) {
          if (S.lane[d] == l) { prod = true; continue; }
          others |= 1u << S.lane[d];
          start = std::max(start, fin[d] + WAIT_MS);
        }
        start += SYNC_MS * (float)__builtin_popcount(others);
        if (lane < 0 || start < best - 1e-6f || (start < best + 1e-6f && prod && !best_prod)) { lane = l; best = start; best_prod = prod; }
      }
      fin[j] = best + std::max(ms[j] - EVENT_MS, 0.002f);
      lane_free[lane] = fin[j] + 0.5f * SYNC_MS;
      S.n_lanes = std::max(S.n_lanes, lane + 1);
    } else {
      // structural heuristic: an op continues the lane of a producer that is still that lane's tail (the producer of
      // in_buf first), otherwise it opens a lane, or takes the one whose tail is oldest
      for (int d : S.deps[j])
        if (lane_tail[S.lane[d]] == d) { lane = S.lane[d]; break; }
      if (lane < 0) {
        if (S.n_lanes < max_lanes) lane = S.n_lanes++;
        else lane = (int)(std::min_element(lane_tail.begin(), lane_tail.end()) - lane_tail.begin());
      }
    }
    S.lane[j] = lane;
    std::vector<int> latest(max_lanes, -1);      // waiting for a lane's latest op covers its earlier ones
    for (int d : S.deps[j])
      if (S.lane[d] != lane && (latest[S.lane[d]] < 0 || pos[d] > pos[latest[S.lane[d]]])) latest[S.lane[d]] = d;
    for (int l = 0; l < max_lanes; ++l)
      if (latest[l] >= 0) { S.wait[j].push_back(latest[l]); S.signal[latest[l]] = 1; }
    lane_tail[lane] = j;
  }
  if (getenv("ACRMI_DEBUG_SCHED")) {
    // depth of the DAG = kernels on the critical path (a single stream runs all of S.order in sequence)
    std::vector<int> depth(n, 0);
    int edges = 0, maxd = 0;
    for (int j : S.order) {
      for (int d : S.deps[j]) depth[j] = std::max(depth[j], depth[d] + 1);
      edges += (int)S.deps[j].size();
      maxd = std::max(maxd, depth[j] + 1);
    }
    int waits = 0;
    for (int j : S.order) waits += (int)S.wait[j].size();
    fprintf(stderr, "[acrmi] schedule(%s, %s batches, %s): %zu ops, %d edges, critical path %d ops; %d lanes, %d cross-lane waits\n",
            point ? "point" : "dense", large ? "large" : "small", planned ? "planned from measured op times" : "structural",
            S.order.size(), edges, maxd, S.n_lanes, waits);
  }
}


extern "C" {

int acrmi_set_program(acrmi_ctx* c, const acrmi_buffer_desc* bufs, int n_bufs, const acrmi_op* ops, int n_ops,
                      const acrmi_head_layout* heads, int max_batch) {
  if (!c || !bufs || !ops || !heads || n_bufs <= 0 || n_ops <= 0 || max_batch <= 0)
    return fail(c, ACRMI_EINVAL, "acrmi_set_program: bad arguments");
  if (!c->weights) return fail(c, ACRMI_ESTATE, "acrmi_set_program: load weights first");
  ON_DEVICE(c);
  // ---- validate before anything is allocated: a malformed program must fail here, not fault on the device
  int prog_dt = ACRMI_DT_F32;      // the one 16-bit storage type of the program, if any
  for (int i = 0; i < n_bufs; ++i) {
    const auto& d = bufs[i];
    if (d.dtype < ACRMI_DT_F32 || d.dtype > ACRMI_DT_BF16) return fail(c, ACRMI_EINVAL, "buffer %d: unknown dtype %d", i, d.dtype);
    if (d.h <= 0 || d.w <= 0 || d.cs <= 0 || d.cs % (d.dtype ? 8 : 4)) return fail(c, ACRMI_EINVAL, "buffer %d: bad geometry", i);
    if (d.dtype) {
      if (prog_dt && prog_dt != d.dtype) return fail(c, ACRMI_EINVAL, "buffer %d: f16 and bf16 buffers in one program", i);
      prog_dt = d.dtype;
    }
  }
  auto bdt = [&](int id) { return id >= 0 ? bufs[id].dtype : (int)ACRMI_DT_F32; };
  auto buf_ok = [&](int id) { return id >= 0 && id < n_bufs; };
  auto w_ok = [&](long long off, long long n) { return off >= 0 && n >= 0 && (unsigned long long)(off + n) <= c->n_weights; };
  {
    const int hb[8] = {heads->center_buf[0], heads->center_buf[1], heads->params_buf[0], heads->params_buf[1],
                       heads->prior_buf[0], heads->prior_buf[1], heads->segm_buf, heads->backbone_buf};
    for (int id : hb)
      if (!buf_ok(id)) return fail(c, ACRMI_EINVAL, "head layout references buffer %d of %d", id, n_bufs);
    for (int k = 0; k < 7; ++k)      // decode / attention pooling / the host read these as fp32 (acr/model.py:56-62 .float())
      if (bufs[hb[k]].dtype != ACRMI_DT_F32) return fail(c, ACRMI_EINVAL, "head layout: head maps must be fp32 buffers");
    if (bufs[heads->params_buf[0]].cs < 109 || bufs[heads->params_buf[1]].cs < 109 || bufs[heads->prior_buf[0]].cs < 106 ||
        bufs[heads->prior_buf[1]].cs < 106)
      return fail(c, ACRMI_EINVAL, "head layout: params/prior buffers are too narrow");
    for (int k = 0; k < 6; ++k)
      if (bufs[hb[k]].h != 64 || bufs[hb[k]].w != 64) return fail(c, ACRMI_EINVAL, "head layout: head maps must be 64x64");
  }
  for (int i = 0; i < n_ops; ++i) {
    const acrmi_op& op = ops[i];
    const int ids[4] = {op.in_buf, op.out_buf, op.res_buf, op.aux_buf};
    for (int id : ids)
      if (id >= n_bufs || id < -1) return fail(c, ACRMI_EINVAL, "op %d references buffer %d of %d", i, id, n_bufs);
    if (op.mode < ACRMI_MODE_BOTH || op.mode > ACRMI_MODE_POINT) return fail(c, ACRMI_EINVAL, "op %d: bad mode", i);
    bool need_in = false, need_out = true;
    switch (op.kind) {
      case ACRMI_OP_U8NORM: case ACRMI_OP_POW11: case ACRMI_OP_COORDFILL: case ACRMI_OP_STEM: break;
      case ACRMI_OP_CONV: case ACRMI_OP_BILINEAR2X: case ACRMI_OP_MAXPOOL: case ACRMI_OP_PAIR1X1: case ACRMI_OP_ATTPOOL: case ACRMI_OP_PAREBIAS: case ACRMI_OP_POINTHEADS:
        need_in = true;
        break;
      case ACRMI_OP_FUSESUM: break;
      default: return fail(c, ACRMI_EINVAL, "op %d: unknown kind %d", i, op.kind);
    }
    if ((need_in && !buf_ok(op.in_buf)) || (need_out && !buf_ok(op.out_buf)))
      return fail(c, ACRMI_EINVAL, "op %d (kind %d): missing input/output buffer", i, op.kind);
    if (op.in_coff < 0 || op.out_coff < 0 || op.res_coff < 0) return fail(c, ACRMI_EINVAL, "op %d: negative channel offset", i);
    if (op.kind == ACRMI_OP_CONV) {
      const int idt = bufs[op.in_buf].dtype, odt = bufs[op.out_buf].dtype;
      if (op.in_coff % (idt ? 8 : 4) || (op.ksize != 1 && op.ksize != 3) || op.stride < 1 || op.stride > 2 || op.cin <= 0 || op.cout <= 0 ||
          op.groups <= 0)
        return fail(c, ACRMI_EINVAL, "op %d: unsupported conv geometry", i);
      const int algo = op.flags & 7;
      if (algo > 7) return fail(c, ACRMI_EINVAL, "op %d: unknown conv algo %d", i, algo);
      // 16-bit input: direct kernel only; the output is 16-bit too or fp32 (a head exit; 1x1 and 3x3 stride 1), a residual
      // has the type of the output.  fp32 input: everything fp32.
      if (idt ? (algo != 0 || (odt != idt && odt != ACRMI_DT_F32) || (odt == ACRMI_DT_F32 && op.stride != 1) ||
                 (op.groups > 1 && op.cin % 2))
              : odt != ACRMI_DT_F32)
        return fail(c, ACRMI_EINVAL, "op %d: conv buffer types do not fit (in %d, out %d, algo %d)", i, idt, odt, algo);
      if (op.res_buf >= 0 && bufs[op.res_buf].dtype != odt)
        return fail(c, ACRMI_EINVAL, "op %d: the residual must have the type of the output", i);
      const bool splitk = (op.flags & ACRMI_CONV_SPLITK) != 0;
      if (splitk && (algo != 2 || idt || op.groups < 2 || op.groups > 8 || op.cin % 32 || op.cin < 64 || op.cout == 33 ||
                     op.bias_per_frame))
        return fail(c, ACRMI_EINVAL, "op %d: split-K needs algo 2, fp32, 2..8 slices of Cin %% 32 == 0, Cin >= 64 channels each", i);
      if (op.flags & ACRMI_CONV_BIAS_MAP) {
        const long long mcs = (op.groups * op.cout + 3) / 4 * 4;
        if (splitk || op.res_buf >= 0 || idt || algo == 3 || !w_ok(op.w_off2, (long long)bufs[op.out_buf].h * bufs[op.out_buf].w * mcs))
          return fail(c, ACRMI_EINVAL, "op %d: a position-bias map needs an fp32 conv without a residual buffer (not algo 3) and "
                      "[Ho][Wo][round4(groups*Cout)] floats inside the blob at w_off2", i);
      }
      const bool dual = (op.flags & ACRMI_CONV_DUAL) != 0;
      if (dual && (algo != 3 || op.nterms < 1 || op.bias_per_frame || !buf_ok(op.aux_buf) || bufs[op.aux_buf].dtype != ACRMI_DT_F32 ||
                   bufs[op.aux_buf].h != bufs[op.out_buf].h || bufs[op.aux_buf].w != bufs[op.out_buf].w || bufs[op.aux_buf].cs % 4 ||
                   bufs[op.aux_buf].cs < op.cout || op.aux_buf == op.out_buf || op.aux_buf == op.in_buf || op.aux_buf == op.res_buf))
        return fail(c, ACRMI_EINVAL, "op %d: a second output needs algo 3, 1..3 terms and an fp32 map of the output's size in aux_buf", i);
      if (op.nterms) {      // extra residual terms (ConvArgs.xt): what conv_pp2_kernel<NT, true> / the 32-cout stride-2 kernel take
        const int ho_ = bufs[op.out_buf].h, wo_ = bufs[op.out_buf].w;
        if (op.nterms < 0 || op.nterms > 3 || idt || (op.flags & ACRMI_CONV_BIAS_MAP) || splitk ||
            (dual ? false : (op.ksize != 3 || op.stride != 2 || (algo != 0 && algo != 5) || op.cout % 32 || op.cin <= 16 ||
                             op.out_coff % 4 || bufs[op.out_buf].cs % 4)))
          return fail(c, ACRMI_EINVAL, "op %d: extra residual terms need an fp32 3x3 stride-2 convolution (algo 0 / 5) with Cin > 16, "
                      "Cout %% 32 = 0 and 16-byte aligned output channels - or ACRMI_CONV_DUAL", i);
        for (int t = 0; t < op.nterms; ++t) {
          const int tb = op.term_buf[t];
          if (!buf_ok(tb) || bufs[tb].dtype != ACRMI_DT_F32 || op.term_coff[t] < 0 || op.term_coff[t] % 4 || bufs[tb].cs % 4 ||
              op.term_shift[t] < 0 || op.term_shift[t] > 3 || op.term_coff[t] + op.groups * op.cout > bufs[tb].cs ||
              (bufs[tb].h << op.term_shift[t]) != ho_ || (bufs[tb].w << op.term_shift[t]) != wo_ || tb == op.out_buf ||
              (dual && tb == op.aux_buf))
            return fail(c, ACRMI_EINVAL, "op %d: residual term %d does not fit the output", i, t);
        }
      }
      if (op.bias_per_frame && buf_ok(op.aux_buf) && bufs[op.aux_buf].dtype != ACRMI_DT_F32)
        return fail(c, ACRMI_EINVAL, "op %d: the per-frame bias must be fp32", i);
      if (algo != 0 && !((algo == 6 || algo == 7) && op.ksize == 3 && op.stride == 2) &&
          !((op.ksize == 3 || (algo >= 6 && op.ksize == 1)) && op.stride == (algo == 5 ? 2 : 1)))
        return fail(c, ACRMI_EINVAL, "op %d: algo %d needs a 3x3 stride-%d convolution", i, algo, algo == 5 ? 2 : 1);
      if (algo == 3 && (op.groups != 1 || op.cin > 32 || op.cout != 32 || op.bias_per_frame || bufs[op.out_buf].h % 8 ||
                        bufs[op.out_buf].w % 16 || op.out_coff % 4 || op.res_coff % 4))
        return fail(c, ACRMI_EINVAL, "op %d: algo 3 needs groups 1, Cin <= 32, Cout = 32, a map of 8x16-pixel tiles", i);
      const int ogroups = splitk ? 1 : op.groups;      // the slices of a split-K conv share the output channels
      if (op.in_coff + op.groups * op.cin > bufs[op.in_buf].cs || op.out_coff + ogroups * op.cout > bufs[op.out_buf].cs ||
          (op.res_buf >= 0 && op.res_coff + ogroups * op.cout > bufs[op.res_buf].cs) ||
          (op.groups > 1 && op.cin % (bufs[op.in_buf].dtype == ACRMI_DT_F32 ? 4 : 8)))      // a group starts on a 16-byte vector
        return fail(c, ACRMI_EINVAL, "op %d: channel slice outside its buffer's channel stride", i);
      const int pad = op.ksize / 2;
      const int ho = (bufs[op.in_buf].h + 2 * pad - op.ksize) / op.stride + 1, wo = (bufs[op.in_buf].w + 2 * pad - op.ksize) / op.stride + 1;
      if (ho != bufs[op.out_buf].h || wo != bufs[op.out_buf].w ||
          (op.res_buf >= 0 && (bufs[op.res_buf].h != ho || bufs[op.res_buf].w != wo)))
        return fail(c, ACRMI_EINVAL, "op %d: output/residual buffer geometry does not match the convolution", i);
      const long long n_tiles = op.cout <= 32 ? 1 : ((op.cout + 63) / 64) * 2;
      if (algo == 4 && (op.cin < 32 || (op.cin == 32 && (op.cout % 32 || ho % 8 || wo % 32))))      // (Cin = 32: conv_wino24b_kernel only)
        return fail(c, ACRMI_EINVAL, "op %d: algo 4 needs Cin > 32, or Cin = 32 with Cout %% 32 = 0 on a map of 8x32-pixel tiles", i);
      if ((algo == 6 || algo == 7) && op.stride == 2 &&
          (op.ksize != 3 || ho % 8 || wo % 32 || bufs[op.in_buf].h != 2 * ho || bufs[op.in_buf].w != 2 * wo || op.nterms > 0))      // (conv_x3s2.inc x3s2_ok)
        return fail(c, ACRMI_EINVAL, "op %d: algo 6 / 7 at stride 2 needs a 3x3 convolution onto a map of 8x32-pixel tiles, even input size, no extra residual terms", i);
      if ((algo == 6 || algo == 7) && (idt || op.cin % 32 || op.cout % 32 || op.out_coff % 4 || op.res_coff % 4 ||
                                       (op.ksize == 3 ? ((ho % 8 || wo % 32) && (ho % 16 || wo % 16)) : ((ho * wo) % 256 != 0))))      // (conv_x3.inc x3_ok / conv_x3p.inc x3p_ok)
        return fail(c, ACRMI_EINVAL, "op %d: algo 6 / 7 needs fp32 storage, Cin %% 32 = 0, Cout %% 32 = 0, a map of 8x32- or 16x16-pixel tiles (3x3) / of whole 256-pixel items (1x1)", i);
      if (algo == 5 && (idt || op.cin % 16 || op.cout % 32 || bufs[op.in_buf].h % 2 || bufs[op.in_buf].w % 2 || ho % 8 || wo % 16 ||
                        op.out_coff % 4 || op.res_coff % 4))      // (conv_pp2.inc pp2_ok)
        return fail(c, ACRMI_EINVAL, "op %d: algo 5 needs fp32, Cin %% 16 = 0, Cout %% 32 = 0, an output map of 8x16-pixel tiles", i);
      const long long taps = algo >= 6 ? op.ksize * op.ksize : algo == 5 ? 28 : algo == 4 ? 24 : (algo >= 2 ? 16 : (algo == 1 ? 12 : op.ksize * op.ksize));
      const long long ksteps = idt ? (op.cin + 15) / 16 : (op.cin + 7) / 8;      // 1 KiB weight fragments per tap and n-tile
      const long long wn = algo == 3 ? 16384 : (long long)op.groups * taps * ksteps * n_tiles * 256 + (algo >= 6 ? 1 : 0);      // (algo 6 / 7: + the weight scale)
      if (!w_ok(op.w_off, wn)) return fail(c, ACRMI_EINVAL, "op %d: packed weights outside the blob", i);
      if (op.bias_per_frame) {
        if (!buf_ok(op.aux_buf) || bufs[op.aux_buf].cs < op.groups * op.cout)
          return fail(c, ACRMI_EINVAL, "op %d: per-frame bias buffer missing or too narrow", i);
      } else if (!w_ok(op.b_off, (long long)op.groups * n_tiles * 32)) {
        return fail(c, ACRMI_EINVAL, "op %d: bias outside the blob", i);
      }
    }
    if (op.kind == ACRMI_OP_STEM) {
      if (bufs[op.out_buf].dtype && (bufs[op.out_buf].cs % 8 || op.out_coff % 8))
        return fail(c, ACRMI_EINVAL, "op %d: a 16-bit stem output needs channel stride / offset in multiples of 8", i);
      if (op.ksize != 3 && op.ksize != 7) return fail(c, ACRMI_EINVAL, "op %d: the stem kernels are 3x3 and 7x7 (stride 2)", i);
      const bool ok = op.ksize == 7 ? stem7_shape_ok(2 * bufs[op.out_buf].h, 2 * bufs[op.out_buf].w, bufs[op.out_buf].cs, op.out_coff)
                                    : stem_shape_ok(2 * bufs[op.out_buf].h, 2 * bufs[op.out_buf].w, bufs[op.out_buf].cs, op.out_coff);
      if (op.cout != 64 || !ok)
        return fail(c, ACRMI_EINVAL, "op %d: the stem kernel needs 64 output channels and a map of 8x64-pixel strips", i);
      if (!w_ok(op.w_off, (op.ksize == 7 ? 74 : 14) * 2 * 64) || !w_ok(op.b_off, 64))
        return fail(c, ACRMI_EINVAL, "op %d: stem weights outside the blob", i);
    }
    if (op.kind == ACRMI_OP_FUSESUM) {
      const int vq = bufs[op.out_buf].dtype ? 8 : 4;      // elements per 16-byte vector
      if (op.nterms < 1 || op.nterms > 4 || op.cout <= 0 || op.cout % vq || op.out_coff % vq || op.out_coff + op.cout > bufs[op.out_buf].cs)
        return fail(c, ACRMI_EINVAL, "op %d: bad fuse-sum geometry", i);
      for (int t = 0; t < op.nterms; ++t) {
        if (!buf_ok(op.term_buf[t]) || op.term_coff[t] < 0 || bufs[op.term_buf[t]].dtype != bufs[op.out_buf].dtype ||
            op.term_coff[t] % vq || op.term_shift[t] < 0 || op.term_shift[t] > 3 ||
            op.term_coff[t] + op.cout > bufs[op.term_buf[t]].cs ||
            (bufs[op.term_buf[t]].h << op.term_shift[t]) != bufs[op.out_buf].h ||
            (bufs[op.term_buf[t]].w << op.term_shift[t]) != bufs[op.out_buf].w)
          return fail(c, ACRMI_EINVAL, "op %d: fuse-sum term %d does not fit the output", i, t);
      }
    }
    if (op.kind == ACRMI_OP_PAIR1X1) {
      if (!buf_ok(op.res_buf) || !buf_ok(op.aux_buf) || op.cin != 64 || op.cout != 256)
        return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair is 64 -> 256 (+ residual) -> 64 with in, res, out and aux buffers", i);
      const int ids4[4] = {op.in_buf, op.res_buf, op.out_buf, op.aux_buf};
      for (int id : ids4)
        if (bufs[id].dtype != ACRMI_DT_F32 || bufs[id].h != bufs[op.in_buf].h || bufs[id].w != bufs[op.in_buf].w)
          return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair's buffers must be fp32 maps of one size", i);
      if (op.in_coff % 4 || op.res_coff % 4 || op.out_coff % 4 || op.in_coff + 64 > bufs[op.in_buf].cs ||
          op.res_coff + 256 > bufs[op.res_buf].cs || op.out_coff + 256 > bufs[op.out_buf].cs || bufs[op.aux_buf].cs < 64 ||
          op.out_buf == op.in_buf || op.aux_buf == op.in_buf || op.aux_buf == op.out_buf || op.aux_buf == op.res_buf)
        return fail(c, ACRMI_EINVAL, "op %d: the 1x1 pair's channel slices do not fit / its buffers alias", i);
      if (!w_ok(op.w_off, PAIR1X1_FLOATS)) return fail(c, ACRMI_EINVAL, "op %d: pair weights outside the blob", i);
    }
    if (op.kind == ACRMI_OP_MAXPOOL) {
      const int vq = bufs[op.in_buf].dtype ? 8 : 4;
      if (op.cin <= 0 || bufs[op.in_buf].dtype != bufs[op.out_buf].dtype || op.cin % vq || op.in_coff % vq || op.out_coff % vq ||
          op.in_coff + op.cin > bufs[op.in_buf].cs || op.out_coff + op.cin > bufs[op.out_buf].cs ||
          bufs[op.out_buf].h != (bufs[op.in_buf].h - 1) / 2 + 1 || bufs[op.out_buf].w != (bufs[op.in_buf].w - 1) / 2 + 1)
        return fail(c, ACRMI_EINVAL, "op %d: bad max-pool geometry", i);
    }
    if (op.kind == ACRMI_OP_BILINEAR2X &&
        (op.cin <= 0 || bufs[op.in_buf].dtype != bufs[op.out_buf].dtype || op.cin % (bufs[op.in_buf].dtype ? 8 : 4) ||
         op.in_coff % (bufs[op.in_buf].dtype ? 8 : 4) || op.out_coff % (bufs[op.in_buf].dtype ? 8 : 4) || op.in_coff + op.cin > bufs[op.in_buf].cs ||
         op.out_coff + op.cin > bufs[op.out_buf].cs || bufs[op.out_buf].h != 2 * bufs[op.in_buf].h ||
         bufs[op.out_buf].w != 2 * bufs[op.in_buf].w))
      return fail(c, ACRMI_EINVAL, "op %d: bad bilinear geometry", i);
    if ((op.kind == ACRMI_OP_POW11 && op.out_coff >= bufs[op.out_buf].cs) ||
        (op.kind == ACRMI_OP_COORDFILL && op.out_coff + 2 > bufs[op.out_buf].cs))
      return fail(c, ACRMI_EINVAL, "op %d: channel outside the buffer", i);
    if (op.kind == ACRMI_OP_ATTPOOL) {
      if (buf_ok(op.res_buf) && (bufs[op.in_buf].dtype != ACRMI_DT_F32 || bufs[op.out_buf].dtype != ACRMI_DT_F32 ||
                                 (bufs[op.res_buf].dtype != ACRMI_DT_F32 && op.cin != 256)))
        return fail(c, ACRMI_EINVAL, "op %d: attention pooling reads fp32 logits, writes fp32, and pools 256 channels of a 16-bit map", i);
      if (!buf_ok(op.res_buf) || (op.cin != 32 && op.cin != 64 && op.cin != 256 && op.cin != 320) ||
          op.res_coff + op.cin > bufs[op.res_buf].cs || bufs[op.in_buf].cs < 33 || bufs[op.in_buf].h != 2 * bufs[op.res_buf].h ||
          bufs[op.in_buf].w != 2 * bufs[op.res_buf].w || (long long)bufs[op.out_buf].h * bufs[op.out_buf].w * bufs[op.out_buf].cs < 32LL * op.cin)
        return fail(c, ACRMI_EINVAL, "op %d: bad attention-pool geometry", i);
    }
    if ((op.kind == ACRMI_OP_PAREBIAS || op.kind == ACRMI_OP_POINTHEADS || op.kind == ACRMI_OP_U8NORM) &&
        (bdt(op.in_buf) || bdt(op.out_buf) || bdt(op.res_buf) || bdt(op.aux_buf)))
      return fail(c, ACRMI_EINVAL, "op %d (kind %d): fp32 buffers only", i, op.kind);
    if (op.kind == ACRMI_OP_PAREBIAS) {
      const long long shape_n = (op.cin == 320 ? 64 : 256) * 16;
      if ((op.cin != 256 && op.cin != 320) || (op.flags != 0 && op.flags != 16) || bufs[op.out_buf].cs < 109 || bufs[op.out_buf].cs > 256 ||
          (long long)bufs[op.in_buf].h * bufs[op.in_buf].w * bufs[op.in_buf].cs < 32LL * op.cin || !w_ok(op.w_off, 6 * 256 * 16) ||
          !w_ok(op.w_off2, 10 * shape_n) || !w_ok(op.b_off2, 10) || !w_ok(op.w_off3, 109 * 106) || !w_ok(op.b_off, 109))
        return fail(c, ACRMI_EINVAL, "op %d: bad pare-bias geometry or weights", i);
    }
    if (op.kind == ACRMI_OP_POINTHEADS) {
      const bool ok = buf_ok(op.res_buf) && buf_ok(op.aux_buf) &&
                      bufs[op.in_buf].h == 128 && bufs[op.in_buf].w == 128 && bufs[op.in_buf].cs == 36 &&
                      bufs[op.res_buf].h == 64 && bufs[op.out_buf].h == 64 && bufs[op.res_buf].cs >= 109 &&
                      bufs[op.out_buf].cs >= 109 && bufs[op.aux_buf].cs >= 109 && op.mode == ACRMI_MODE_POINT &&
                      w_ok(op.w_off, 3LL * TP_TOWER_FLOATS) && w_ok(op.w_off2, 109LL * TP_EXIT_N);
      if (!ok) return fail(c, ACRMI_EINVAL, "op %d: unsupported point-heads geometry", i);
    }
  }
  free_program(c);
  c->bufs.assign(bufs, bufs + n_bufs);
  c->ops.assign(ops, ops + n_ops);
  c->heads = *heads;
  c->max_batch = max_batch;
  c->buf_ptr.assign(n_bufs, nullptr);
  for (int i = 0; i < n_bufs; ++i) {
    const auto& d = bufs[i];
    const size_t bytes = (size_t)max_batch * d.h * d.w * d.cs * (d.dtype ? 2 : sizeof(float));
    hipError_t e = hipMalloc(&c->buf_ptr[i], bytes);
    if (e != hipSuccess) return fail(c, ACRMI_ENOMEM, "hipMalloc(%zu) for buffer %d: %s", bytes, i, hipGetErrorString(e));
    HIPCHK(c, hipMemset(c->buf_ptr[i], 0, bytes));
  }
  for (int i = 0; i < n_ops; ++i) {
    const acrmi_op& op = ops[i];
    if (op.kind != ACRMI_OP_CONV || !(op.flags & ACRMI_CONV_SPLITK)) continue;
    const auto& d = bufs[op.out_buf];
    c->split_ws_floats = std::max(c->split_ws_floats, conv_splitk_ws_floats(max_batch, d.h, d.w, op.cout, op.groups));
    c->split_counters = std::max(c->split_counters, conv_splitk_counters(max_batch, d.h, d.w, op.cout));
  }
  if (c->split_ws_floats)
    for (int l = 0; l < MAX_LANES; ++l) {
      HIPCHK(c, hipMalloc(&c->split_ws[l], c->split_ws_floats * sizeof(float)));
      HIPCHK(c, hipMalloc(&c->split_cnt[l], c->split_counters * sizeof(unsigned)));
      HIPCHK(c, hipMemset(c->split_cnt[l], 0, c->split_counters * sizeof(unsigned)));
    }
  c->att_ws_floats = attpool_ws_floats(max_batch, 320);
  HIPCHK(c, hipMalloc(&c->att_ws, c->att_ws_floats * sizeof(float)));
  HIPCHK(c, hipMalloc(&c->picks, (size_t)max_batch * 4 * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->gate_buf, (size_t)max_batch * sizeof(int)));
  for (int i = 0; i < n_ops; ++i)
    if (ops[i].kind == ACRMI_OP_CONV && (ops[i].flags & 7) == 6 && !c->range_flag) {
      HIPCHK(c, hipMalloc(&c->range_flag, sizeof(unsigned)));
      HIPCHK(c, hipMemset(c->range_flag, 0, sizeof(unsigned)));
    }
  c->op_ms[0].clear(); c->op_ms[1].clear();      // measured times belong to the previous program
  for (int v = 0; v < 4; ++v) build_schedule(c, v & 1, v & 2);
  c->op_ev.assign(n_ops, nullptr);
  c->have_program = true;
  // init-time ops (constants that live in persistent buffers)
  for (const acrmi_op& op : c->ops)
    if (op.kind == ACRMI_OP_COORDFILL) {
      int r = run_op(c, op, nullptr, max_batch, nullptr);
      if (r) return r;
    }
  HIPCHK(c, hipDeviceSynchronize());
  return ACRMI_OK;
}

}  // extern "C"

// The program with its independent chains on parallel streams: lane 0 is the caller's stream, the other lanes fork
// from it (so they start after everything queued before this call) and join it at the end.
static unsigned lane_event_flags() {
  static const unsigned f = [] {
    const char* e = experiment_env("ACRMI_EVENT_FLAGS");      // experiment switch: extra hipEventCreateWithFlags bits (hex)
    if (e) fprintf(stderr, "[acrmi] WARNING: ACRMI_EVENT_FLAGS=%s - lane events created with non-default flags (experiment)\n", e);
    return hipEventDisableTiming | (e ? (unsigned)strtoul(e, nullptr, 16) : 0u);
  }();
  return f;
}

static int run_program_lanes(acrmi_ctx* c, const uint8_t* img, int B, hipStream_t user, bool point) {
  const Schedule& S = c->sched[point ? 1 : 0][B > AUTO_SMALL_BATCH ? 1 : 0];
  for (int l = 1; l < S.n_lanes; ++l) {
    if (!c->lanes[l]) HIPCHK(c, hipStreamCreateWithFlags(&c->lanes[l], hipStreamNonBlocking));
    if (!c->join_ev[l]) HIPCHK(c, hipEventCreateWithFlags(&c->join_ev[l], lane_event_flags()));
  }
  if (!c->fork_ev) HIPCHK(c, hipEventCreateWithFlags(&c->fork_ev, lane_event_flags()));
  auto st = [&](int l) { return l == 0 ? user : c->lanes[l]; };
  HIPCHK(c, hipEventRecord(c->fork_ev, user));
  for (int l = 1; l < S.n_lanes; ++l) HIPCHK(c, hipStreamWaitEvent(c->lanes[l], c->fork_ev, 0));
  int r = ACRMI_OK;
  for (int j : S.order) {
    hipStream_t s = st(S.lane[j]);
    // timing ablation (WRONG results: data races between lanes): 1 = no waits, 2 = no waits and no records.  Loud, so that a
    // leaked environment variable cannot silently corrupt a real run.
    static const int ablate = [] {
      const char* e = experiment_env("ACRMI_ABLATE_LANE_SYNC");
      const int v = e ? atoi(e) : 0;
      if (v) fprintf(stderr, "[acrmi] WARNING: ACRMI_ABLATE_LANE_SYNC=%d - cross-lane synchronisation is DISABLED, results are WRONG "
                             "(timing experiment only)\n", v);
      return v;
    }();
    if (!(ablate & 1)) for (int d : S.wait[j]) HIPCHK(c, hipStreamWaitEvent(s, c->op_ev[d], 0));
    r = run_op(c, c->ops[j], img, B, s, S.lane[j]);
    if (r) break;
    if (S.signal[j] && !(ablate & 2)) {
      if (!c->op_ev[j]) HIPCHK(c, hipEventCreateWithFlags(&c->op_ev[j], lane_event_flags()));
      HIPCHK(c, hipEventRecord(c->op_ev[j], s));
    }
  }
  for (int l = 1; l < S.n_lanes; ++l) {       // join also on the error path: nothing may outlive the call unordered
    HIPCHK(c, hipEventRecord(c->join_ev[l], c->lanes[l]));
    HIPCHK(c, hipStreamWaitEvent(user, c->join_ev[l], 0));
  }
  return r;
}

// first_op > 0 (acrmi_heads): only ops [first_op, n) run - in program order on the caller's stream (the lanes' events of the
// skipped ops would be stale) - and `img` is not read
int run_program(acrmi_ctx* c, const uint8_t* img, int B, void* stream, bool point, int first_op) {
  if (!c || (!img && first_op <= 0)) return fail(c, ACRMI_EINVAL, "acrmi_backbone_heads: bad arguments");
  if (!c->have_program) return fail(c, ACRMI_ESTATE, "acrmi_backbone_heads: no program");
  if (B <= 0 || B > c->max_batch) return fail(c, ACRMI_EINVAL, "batch %d outside 1..%d", B, c->max_batch);
  ON_DEVICE(c);
  static const bool dbg_sync = getenv("ACRMI_DEBUG_SYNC") != nullptr;   // attribute a fault/hang to an op
  if (c->sched[point ? 1 : 0][B > AUTO_SMALL_BATCH ? 1 : 0].n_lanes > 1 && !dbg_sync && first_op <= 0)
    return run_program_lanes(c, img, B, (hipStream_t)stream, point);
  int i = 0;
  for (const acrmi_op& op : c->ops) {
    ++i;
    if (!op_active(op, point) || i - 1 < first_op) continue;
    if (dbg_sync) fprintf(stderr, "[acrmi] op %d kind %d B %d\n", i - 1, (int)op.kind, B), fflush(stderr);
    int r = run_op(c, op, img, B, (hipStream_t)stream);
    if (r) return r;
    if (dbg_sync) HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
  }
  return ACRMI_OK;
}

