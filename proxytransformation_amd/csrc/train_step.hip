// The training step of the module as ONE enqueue call per direction (include/proxyt.h "the whole training step as ONE call per
// direction"; VERDICT r05 "next" #4).
//
// Nothing here is a kernel: the two entry points chain the stage / operator / composite entry points of this library (the same
// ones proxytransformation_amd/train.py's one-node step called one ctypes call at a time) from C++, with every intermediate carved
// out of two caller-owned arenas.  r05's host anatomy of a step (scratch/train_hostprof6.py): 0.49 ms inside the library for 35 calls,
// 0.14 ms in 60 torch.empty, ~0.6 ms of Python in the two bodies around them -- on a box whose host is the limit the step ran at the
// host's pace (2.0 - 2.3 ms against 1.7 ms of GPU time).  The launches, the two streams and their events are those of train.py's
// _TrainStep (image branch on the side stream, the image block beside the text block, the backward mirrored); the ORDER in which the
// two streams are fed follows the kernel timeline of a step (profiles/r06_train_timeline.txt): see the comments at the two places.
#include <hip/hip_runtime.h>

#include "common.h"

using namespace ptx;

namespace {

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t r = off; off += align_up(bytes > 0 ? bytes : 1, 256); return r; }
};

struct StepLay {
    // forward arena (byte offsets)
    size_t minmax, c0, enc_scratch, idx1, cl1, pooled, mr_off, st_off, sc_off, raw, centers, dcoef, idx2, cl2, pad, order, picks, keep,
        kcenter_i, kcluster, kidx, drop_idx, tag, tile_counts, opos, counts, src, kcenter, pp, arg_enc, mr_enc, st_enc, sc_enc,
        ip_save, ip_tmp, img_proxy, tb_save, tb_tmp, translate, ib_save, ib_tmp, transform, fwd_total;
    size_t sc_off_bytes, sc_enc_bytes, ip_save_f, ip_tmp_f, ip_tmpb_f, tb_save_f, tb_tmp_f, tb_tmpb_f, ib_save_f, ib_tmp_f, ib_tmpb_f;
    // backward arena
    size_t dt, dT, dc, dx_i, dproxy_i, ib_tmpb, ip_tmpb, dx_t, dproxy_t, tb_tmpb, dpp, dkc_enc, dkcenter, dcenters, dpooled, hb_tmp,
        sc_bwd, zeros, bwd_total;
    size_t hb_tmp_f;
    int64_t goff[PTX_TS_NGRAD];
    size_t grads_floats;
    int M, Kd, ntiles;
};

int block_numel(const PtxTrainBlock &b, int i, long &n)
{
    const long C = b.C, H = b.H, nn = b.n, s = b.s, no = b.nout;
    switch (i) {
    case PTX_TB_LN1_W: case PTX_TB_LN1_B: case PTX_TB_LN2_W: case PTX_TB_LN2_B: case PTX_TB_LN3_W: case PTX_TB_LN3_B:
    case PTX_TB_PP_B: case PTX_TB_PROJ_B: case PTX_TB_FC2_B: n = C; break;
    case PTX_TB_PB: n = 16 * nn; break;
    case PTX_TB_PC: case PTX_TB_PR: n = nn * s; break;
    case PTX_TB_QKV_W: n = 3 * C * C; break;
    case PTX_TB_QKV_B: n = 3 * C; break;
    case PTX_TB_PP_W: case PTX_TB_PROJ_W: n = C * C; break;
    case PTX_TB_FC1_W: case PTX_TB_FC2_W: n = H * C; break;
    case PTX_TB_FC1_B: n = H; break;
    case PTX_TB_HEAD_W: n = no * C; break;
    case PTX_TB_HEAD_B: case PTX_TB_BN_W: case PTX_TB_BN_B: n = no; break;
    default: return PTX_EINVAL;
    }
    return PTX_OK;
}

int step_layout(const PtxTrainStep &a, StepLay &L)
{
    const PtxShape &s = a.shape;
    PTX_REQUIRE(s.B >= 1 && s.B <= 32 && s.N >= 1 && s.grid_size >= 1 && s.K >= 1 && s.Mk >= 1 && s.Mt >= s.Mk,
                "ptx_train_step: bad shape (B=%d N=%d gs=%d K=%d Mt=%d Mk=%d)", s.B, s.N, s.grid_size, s.K, s.Mt, s.Mk);
    const int B = s.B, N = s.N, K = s.K, Mk = s.Mk, Mt = s.Mt, C = s.C;
    const long M = (long)s.grid_size * s.grid_size * s.grid_size;
    L.M = (int)M; L.Kd = Mt - Mk; L.ntiles = cdiv(N, kTilePts);
    const long BM = (long)B * M, R = (long)B * Mk, Kd1 = L.Kd > 1 ? L.Kd : 1;
    const int Wo = a.off.W, We = a.enc.W;
    PTX_REQUIRE(Wo >= 64 && Wo % 64 == 0 && Wo <= 512 && We == C, "ptx_train_step: slot network widths %d / %d (embed_dim %d)", Wo, We, C);
    PTX_REQUIRE(a.tb.B == B && a.ib.B == B && a.tb.n == Mk && a.ib.n == Mk && a.tb.C == C && a.ib.C == C && a.tb.L == s.L &&
                a.ib.L == s.V && a.tb.nout == 3 && a.ib.nout == 9 && a.ip.nimg == B * s.V && a.ip.C == C,
                "ptx_train_step: the block / image-pool descriptions do not match the shape");
    size_t f0, f1, f2;
    PTX_TRY(ptx_train_imgpool_sizes(&a.ip, &f0, &f1, &f2));
    L.ip_save_f = f0; L.ip_tmp_f = f1; L.ip_tmpb_f = f2;
    PTX_TRY(ptx_train_block_sizes(&a.tb, &f0, &f1, &f2));
    L.tb_save_f = f0; L.tb_tmp_f = f1; L.tb_tmpb_f = f2;
    PTX_TRY(ptx_train_block_sizes(&a.ib, &f0, &f1, &f2));
    L.ib_save_f = f0; L.ib_tmp_f = f1; L.ib_tmpb_f = f2;
    L.sc_off_bytes = ptx_op_slotnet_scratch_bytes(Wo);
    L.sc_enc_bytes = ptx_op_slotnet_scratch_bytes(We);
    Carver c;
    L.minmax = c.take(B * 6 * 4); L.c0 = c.take(BM * 3 * 4); L.enc_scratch = c.take((B * 6 > 64 ? B * 6 : 64) * 4);
    L.idx1 = c.take(BM * K * 4); L.cl1 = c.take(BM * K * 3 * 4);
    L.pooled = c.take(BM * Wo * 4); L.mr_off = c.take(2 * Wo * 4); L.st_off = c.take(2 * Wo * 4); L.sc_off = c.take(L.sc_off_bytes);
    L.raw = c.take(BM * 3 * 4); L.centers = c.take(BM * 3 * 4); L.dcoef = c.take(BM * 3 * 4);
    L.idx2 = c.take(BM * K * 4); L.cl2 = c.take(BM * K * 3 * 4); L.pad = c.take(BM * 4);
    L.order = c.take((long)B * Mt * 4); L.picks = c.take(B * Kd1 * 4); L.keep = c.take(R * 4);
    L.kcenter_i = c.take(R * 3 * 4); L.kcluster = c.take(R * K * 3 * 4); L.kidx = c.take(R * K * 4);
    L.drop_idx = c.take(B * Kd1 * K * 4); L.tag = c.take((long)B * N * 4);
    L.tile_counts = c.take((long)B * L.ntiles * 4); L.opos = c.take((long)B * N * 4); L.counts = c.take(B * 4); L.src = c.take(R * 4);
    L.kcenter = c.take(R * 3 * 4); L.pp = c.take(R * We * 4); L.arg_enc = c.take(R * We * 4);
    L.mr_enc = c.take(2 * We * 4); L.st_enc = c.take(2 * We * 4); L.sc_enc = c.take(L.sc_enc_bytes);
    L.ip_save = c.take(L.ip_save_f * 4); L.ip_tmp = c.take(L.ip_tmp_f * 4); L.img_proxy = c.take((long)B * s.V * C * 4);
    L.tb_save = c.take(L.tb_save_f * 4); L.tb_tmp = c.take(L.tb_tmp_f * 4); L.translate = c.take(R * 3 * 4);
    L.ib_save = c.take(L.ib_save_f * 4); L.ib_tmp = c.take(L.ib_tmp_f * 4); L.transform = c.take(R * 9 * 4);
    L.fwd_total = c.off;
    Carver d;
    L.dt = d.take(R * 3 * 4); L.dT = d.take(R * 9 * 4); L.dc = d.take(R * 3 * 4);
    L.dx_i = d.take(R * C * 4); L.dproxy_i = d.take((long)B * s.V * C * 4); L.ib_tmpb = d.take(L.ib_tmpb_f * 4);
    L.ip_tmpb = d.take(L.ip_tmpb_f * 4);
    L.dx_t = d.take(R * C * 4); L.dproxy_t = d.take((long)B * s.L * C * 4); L.tb_tmpb = d.take(L.tb_tmpb_f * 4);
    L.dpp = d.take(R * C * 4); L.dkc_enc = d.take(R * 3 * 4); L.dkcenter = d.take(R * 3 * 4); L.dcenters = d.take(BM * 3 * 4);
    L.dpooled = d.take(BM * Wo * 4);
    L.hb_tmp_f = ptx_op_head_bwd_tmp_floats((int)BM, Wo, 3);
    L.hb_tmp = d.take(L.hb_tmp_f * 4);
    L.sc_bwd = d.take(L.sc_off_bytes > L.sc_enc_bytes ? L.sc_off_bytes : L.sc_enc_bytes);
    L.zeros = d.take((long)N * 3 * 4);                 // a scene whose output gradient did not arrive (None) reads zeros
    L.bwd_total = d.off;
    // gradients: every region starts on a 64-float boundary
    long g = 0;
    auto put = [&](int slot, long numel) { L.goff[slot] = g; g += (numel + 63) / 64 * 64; };
    put(PTX_TS_OFF_CONV_W, (long)Wo * 6); put(PTX_TS_OFF_CONV_B, Wo);
    L.goff[PTX_TS_OFF_BN_B] = g; L.goff[PTX_TS_OFF_BN_W] = g + Wo; g += (2 * Wo + 63) / 64 * 64;      // ptx_op_slotnet_bwd: (2,W) = [dbeta | dgamma]
    put(PTX_TS_MAP_W, 3L * Wo);
    put(PTX_TS_ENC_CONV_W, (long)We * 6); put(PTX_TS_ENC_CONV_B, We);
    L.goff[PTX_TS_ENC_BN_B] = g; L.goff[PTX_TS_ENC_BN_W] = g + We; g += (2 * We + 63) / 64 * 64;
    const long Cin = a.ip.Cin, hw = a.ip.hw;
    const long ipn[13] = {C * Cin, C, (hw + 1) * C, (long)C * C, C, (long)C * C, C, (long)C * C, C, (long)C * C, C, C, C};
    const bool tail = a.ip.cw != nullptr;
    for (int i = 0; i < 13; ++i) {
        if (i >= 9 && !tail) L.goff[PTX_TS_IP0 + i] = -1;
        else put(PTX_TS_IP0 + i, ipn[i]);
    }
    for (int blk = 0; blk < 2; ++blk) {
        const PtxTrainBlock &b = blk == 0 ? a.tb : a.ib;
        const int base = blk == 0 ? PTX_TS_TB0 : PTX_TS_IB0;
        for (int i = 0; i < PTX_TB_NPARAM; ++i) {
            if (b.param[i] == nullptr) { L.goff[base + i] = -1; continue; }
            long n;
            PTX_TRY(block_numel(b, i, n));
            put(base + i, n);
        }
    }
    L.grads_floats = (size_t)g;
    return PTX_OK;
}

template <typename T>
T *at(void *base, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(base) + off); }

int record_wait(hipEvent_t ev, hipStream_t from, hipStream_t to)
{
    PTX_HIP(hipEventRecord(ev, from));
    PTX_HIP(hipStreamWaitEvent(to, ev, 0));
    return PTX_OK;
}

float *gptr(const PtxTrainStep &a, const StepLay &L, int slot) { return L.goff[slot] < 0 ? nullptr : a.grads + L.goff[slot]; }

}  // namespace

extern "C" {

int ptx_train_step_layout(const PtxTrainStep *a, PtxTrainStepLayout *out)
{
    PTX_REQUIRE(a && out, "ptx_train_step_layout: null argument");
    StepLay L{};
    PTX_TRY(step_layout(*a, L));
    out->arena_fwd_bytes = L.fwd_total; out->arena_bwd_bytes = L.bwd_total; out->grads_floats = L.grads_floats;
    out->idx2 = L.idx2; out->order = L.order; out->picks = L.picks; out->keep = L.keep; out->kidx = L.kidx; out->drop_idx = L.drop_idx;
    out->centers = L.centers; out->translate = L.translate; out->transform = L.transform; out->point_proxy = L.pp;
    out->img_proxy = L.img_proxy; out->kcenter = L.kcenter; out->opos = L.opos;
    for (int i = 0; i < PTX_TS_NGRAD; ++i) out->grad_off[i] = L.goff[i];
    return PTX_OK;
}

int ptx_train_step_fwd(const PtxTrainStep *ap, void *stream)
{
    PTX_REQUIRE(ap && ap->points && ap->lin && ap->text_feats && ap->out && ap->counts_host && ap->arena_fwd && ap->ws && ap->map_w,
                "ptx_train_step_fwd: null argument");
    const PtxTrainStep &a = *ap;
    StepLay L{};
    PTX_TRY(step_layout(a, L));
    PTX_REQUIRE(a.arena_fwd_bytes >= L.fwd_total && (reinterpret_cast<uintptr_t>(a.arena_fwd) & 255) == 0,
                "ptx_train_step_fwd: forward arena too small (%zu < %zu bytes) or not 256-byte aligned", a.arena_fwd_bytes, L.fwd_total);
    PTX_REQUIRE(a.side_stream == nullptr || (a.ev_fork && a.ev_join && a.ev_pp), "ptx_train_step_fwd: a side stream needs its events");
    PTX_REQUIRE(a.ev_counts != nullptr, "ptx_train_step_fwd: ev_counts");
    const PtxShape &s = a.shape;
    const int B = s.B, N = s.N, K = s.K, Mk = s.Mk, Mt = s.Mt, M = L.M;
    const long BM = (long)B * M, R = (long)B * Mk;
    hipStream_t main = static_cast<hipStream_t>(stream), side = static_cast<hipStream_t>(a.side_stream);
    hipEvent_t ev_fork = static_cast<hipEvent_t>(a.ev_fork), ev_join = static_cast<hipEvent_t>(a.ev_join),
               ev_pp = static_cast<hipEvent_t>(a.ev_pp), ev_counts = static_cast<hipEvent_t>(a.ev_counts);
    void *A = a.arena_fwd;
    // ---- enqueue ORDER (r06, from the kernel timeline of a step, profiles/r06_train_timeline.txt): the caller's stream carries the
    // chain the forward waits for (both ball queries, 0.2 ms of sequential farthest-point picks, the slot networks), so its first
    // dozen launches go out first; the image branch (PRE:449-450: it needs nothing from the clustering half and is ready 0.3 ms
    // before its consumer) follows on the side stream, and the output positions + the early copy of the list lengths -- needed by
    // the affine apply and the backward only -- run there too, off the chain
    PtxTrainImgPool ip = a.ip;
    ip.save = at<float>(A, L.ip_save); ip.save_floats = L.ip_save_f; ip.tmp = at<float>(A, L.ip_tmp); ip.tmp_floats = L.ip_tmp_f;
    if (ip.cw != nullptr) { ip.proxy = at<float>(A, L.img_proxy); ip.o = nullptr; } else { ip.o = at<float>(A, L.img_proxy); }
    if (side) PTX_TRY(record_wait(ev_fork, main, side));                // everything the caller queued before the step
    // ---- index half, part 1 + offset network (PRE:55-62)
    float *c0 = at<float>(A, L.c0), *minmax = at<float>(A, L.minmax);
    PTX_TRY(ptx_grid_centers(a.points, B, N, a.lin, s.grid_size, s.margin, minmax, c0, at<void>(A, L.enc_scratch),
                             (size_t)(B * 6 > 64 ? B * 6 : 64) * 4, main));
    PTX_TRY(ptx_ball_query(c0, a.points, B, M, N, K, s.radius, at<int32_t>(A, L.idx1), at<float>(A, L.cl1), nullptr, main));
    PTX_TRY(ptx_op_slotnet_fwd(c0, at<float>(A, L.cl1), BM, K, a.off.W, a.off.conv_w, a.off.conv_b, a.off.bn_w, a.off.bn_b, a.off.eps,
                               a.off.momentum, a.off.run_mean, a.off.run_var, 0, at<float>(A, L.pooled), nullptr, at<float>(A, L.mr_off),
                               at<float>(A, L.st_off), at<void>(A, L.sc_off), L.sc_off_bytes, main));
    // OffsetHead: raw = pooled map_w^T, then tanh * margin + grid centre, clamped (PRE:59-62, 103)
    PTX_TRY(ptx_op_gemm(at<float>(A, L.pooled), a.map_w, at<float>(A, L.raw), (int)BM, 3, a.off.W, a.off.W, 1, 1, a.off.W, 3, 1, 1, 1, 0, 0,
                        0, 0, 0, 0, 0, 0, 1.0f, 0, 1, 0, main));
    float *centers = at<float>(A, L.centers);
    PTX_TRY(ptx_op_offset_apply(c0, at<float>(A, L.raw), minmax, BM, M, s.margin, centers, at<float>(A, L.dcoef), main));
    // ---- index half, part 2 (PRE:65, 352-420, 478-523)
    const float *cdet = a.centers_override ? a.centers_override : centers;
    PTX_TRY(ptx_ball_query(cdet, a.points, B, M, N, K, s.radius, at<int32_t>(A, L.idx2), at<float>(A, L.cl2), at<int32_t>(A, L.pad), main));
    PTX_TRY(ptx_select_clusters(&s, at<int32_t>(A, L.idx2), cdet, at<float>(A, L.cl2), at<int32_t>(A, L.pad), a.order_override,
                                at<int32_t>(A, L.order), at<int32_t>(A, L.picks), at<int32_t>(A, L.keep), at<float>(A, L.kcenter_i),
                                at<float>(A, L.kcluster), at<int32_t>(A, L.kidx), at<int32_t>(A, L.drop_idx), at<uint32_t>(A, L.tag), main));
    // ---- side stream: image branch, then (behind the selection's tags) output positions + the list lengths of PRE:467, copied out
    // now and awaited by the host after everything is enqueued
    hipStream_t aux = side ? side : main;
    PTX_TRY(ptx_train_imgpool_fwd(&ip, aux));
    if (side) PTX_TRY(record_wait(ev_pp, main, side));                   // the tags are final
    PTX_TRY(ptx_op_out_positions(at<uint32_t>(A, L.tag), B, N, at<int32_t>(A, L.tile_counts), at<int32_t>(A, L.opos),
                                 at<int32_t>(A, L.counts), aux));
    PTX_HIP(hipMemcpyAsync(a.counts_host, at<int32_t>(A, L.counts), (size_t)B * 4, hipMemcpyDeviceToHost, aux));
    PTX_HIP(hipEventRecord(ev_counts, aux));
    PTX_TRY(ptx_op_keep_rows(at<int32_t>(A, L.order), at<int32_t>(A, L.keep), B, M, Mt, Mk, at<int32_t>(A, L.src), main));
    // ---- float half (PRE:437-455)
    float *kcenter = at<float>(A, L.kcenter), *pp = at<float>(A, L.pp);
    PTX_TRY(ptx_op_rows_gather(centers, at<int32_t>(A, L.src), R, 3, kcenter, main));
    PTX_TRY(ptx_op_slotnet_fwd(kcenter, at<float>(A, L.kcluster), R, K, a.enc.W, a.enc.conv_w, a.enc.conv_b, a.enc.bn_w, a.enc.bn_b,
                               a.enc.eps, a.enc.momentum, a.enc.run_mean, a.enc.run_var, 1, pp, at<int32_t>(A, L.arg_enc),
                               at<float>(A, L.mr_enc), at<float>(A, L.st_enc), at<void>(A, L.sc_enc), L.sc_enc_bytes, main));
    PtxTrainBlock tb = a.tb, ib = a.ib;
    tb.x = pp; tb.proxy = a.text_feats; tb.mask = a.text_mask; tb.out = at<float>(A, L.translate);
    tb.save = at<float>(A, L.tb_save); tb.save_floats = L.tb_save_f; tb.tmp = at<float>(A, L.tb_tmp); tb.tmp_floats = L.tb_tmp_f;
    ib.x = pp; ib.proxy = at<float>(A, L.img_proxy); ib.mask = nullptr; ib.out = at<float>(A, L.transform);
    ib.save = at<float>(A, L.ib_save); ib.save_floats = L.ib_save_f; ib.tmp = at<float>(A, L.ib_tmp); ib.tmp_floats = L.ib_tmp_f;
    const bool apart = side != nullptr && a.blocks_apart != 0;
    if (apart) {
        // the image block follows its pooling pass on the side stream, beside the text block on the caller's stream
        PTX_TRY(record_wait(ev_pp, main, side));
        PTX_TRY(ptx_train_block_fwd(&ib, side));
    }
    PTX_TRY(ptx_train_block_fwd(&tb, main));
    if (side) PTX_TRY(record_wait(ev_join, side, main));
    if (!apart) PTX_TRY(ptx_train_block_fwd(&ib, main));
    // ---- submanifold reshape + scatter + drop (PRE:459-467)
    PTX_TRY(ptx_affine_compact(&s, a.points, at<uint32_t>(A, L.tag), kcenter, at<float>(A, L.translate), at<float>(A, L.transform), a.out,
                               at<int32_t>(A, L.counts), a.ws, a.ws_bytes, main));
    return PTX_OK;
}

int ptx_train_step_bwd(const PtxTrainStep *ap, void *stream)
{
    PTX_REQUIRE(ap && ap->douts && ap->arena_fwd && ap->arena_bwd && ap->grads, "ptx_train_step_bwd: null argument");
    const PtxTrainStep &a = *ap;
    StepLay L{};
    PTX_TRY(step_layout(a, L));
    PTX_REQUIRE(a.arena_bwd_bytes >= L.bwd_total && a.grads_floats >= L.grads_floats &&
                ((reinterpret_cast<uintptr_t>(a.arena_bwd) | reinterpret_cast<uintptr_t>(a.grads)) & 255) == 0,
                "ptx_train_step_bwd: backward arena / gradient buffer too small or not 256-byte aligned");
    const PtxShape &s = a.shape;
    const int B = s.B, N = s.N, K = s.K, Mk = s.Mk, M = L.M, C = s.C;
    const long BM = (long)B * M, R = (long)B * Mk;
    hipStream_t main = static_cast<hipStream_t>(stream), side = static_cast<hipStream_t>(a.side_stream);
    hipEvent_t ev_fork = static_cast<hipEvent_t>(a.ev_fork), ev_join = static_cast<hipEvent_t>(a.ev_join), ev_pp = static_cast<hipEvent_t>(a.ev_pp);
    void *A = a.arena_fwd, *D = a.arena_bwd;
    // ---- affine apply: one output gradient per scene (a missing one reads zeros)
    const float *dptr[32];
    bool need_zero = false;
    for (int b = 0; b < B; ++b) { dptr[b] = a.douts[b] ? a.douts[b] : at<float>(D, L.zeros); need_zero |= a.douts[b] == nullptr; }
    if (need_zero) PTX_HIP(hipMemsetAsync(at<void>(D, L.zeros), 0, (size_t)N * 3 * 4, main));
    float *dt = at<float>(D, L.dt), *dT = at<float>(D, L.dT), *dc = at<float>(D, L.dc);
    PTX_TRY(ptx_op_affine_bwd_list(dptr, at<int32_t>(A, L.opos), at<int32_t>(A, L.kidx), at<float>(A, L.kcluster), at<float>(A, L.kcenter),
                                   at<float>(A, L.transform), B, N, Mk, K, dt, dT, dc, main));
    // gradients that arrive through the per-cluster transforms handed out by return_transforms
    if (a.g_kcenter) PTX_TRY(ptx_op_eltwise(0, dc, a.g_kcenter, 0.0f, R * 3, 1, dc, main));
    if (a.g_translate) PTX_TRY(ptx_op_eltwise(0, dt, a.g_translate, 0.0f, R * 3, 1, dt, main));
    if (a.g_transform) PTX_TRY(ptx_op_eltwise(0, dT, a.g_transform, 0.0f, R * 9, 1, dT, main));
    PtxTrainBlock tb = a.tb, ib = a.ib;
    float *pp = at<float>(A, L.pp);
    tb.x = pp; tb.proxy = a.text_feats; tb.mask = a.text_mask; tb.out = at<float>(A, L.translate);
    tb.save = at<float>(A, L.tb_save); tb.save_floats = L.tb_save_f; tb.tmp = at<float>(D, L.tb_tmpb); tb.tmp_floats = L.tb_tmpb_f;
    tb.dout = dt; tb.dx = at<float>(D, L.dx_t); tb.dproxy = a.dtext ? a.dtext : at<float>(D, L.dproxy_t); tb.dx_add = nullptr;
    ib.x = pp; ib.proxy = at<float>(A, L.img_proxy); ib.mask = nullptr; ib.out = at<float>(A, L.transform);
    ib.save = at<float>(A, L.ib_save); ib.save_floats = L.ib_save_f; ib.tmp = at<float>(D, L.ib_tmpb); ib.tmp_floats = L.ib_tmpb_f;
    ib.dout = dT; ib.dx = at<float>(D, L.dx_i); ib.dproxy = at<float>(D, L.dproxy_i); ib.dx_add = nullptr;
    for (int i = 0; i < PTX_TB_NPARAM; ++i) { tb.grad[i] = gptr(a, L, PTX_TS_TB0 + i); ib.grad[i] = gptr(a, L, PTX_TS_IB0 + i); }
    PtxTrainImgPool ip = a.ip;
    ip.save = at<float>(A, L.ip_save); ip.save_floats = L.ip_save_f; ip.tmp = at<float>(D, L.ip_tmpb); ip.tmp_floats = L.ip_tmpb_f;
    ip.dimg = a.dimg;
    float **ipg[13] = {&ip.dwc, &ip.dbc, &ip.dpos, &ip.dwq, &ip.dbq, &ip.dwk, &ip.dbk, &ip.dwv, &ip.dbv, &ip.dcw, &ip.dcb, &ip.dlnw, &ip.dlnb};
    for (int i = 0; i < 13; ++i) *ipg[i] = gptr(a, L, PTX_TS_IP0 + i);
    if (ip.cw != nullptr) { ip.proxy = at<float>(A, L.img_proxy); ip.dproxy = ib.dproxy; } else { ip.o = at<float>(A, L.img_proxy); ip.dout = ib.dproxy; }
    const bool apart = side != nullptr && a.blocks_apart != 0;
    float *dpp;
    if (apart) {
        // image block AND image branch beside the text block's backward on the caller's stream
        // (enqueue order, r06: the text block's backward goes out BEFORE the image pool's -- enqueued behind the side stream's ~55
        //  launches the caller's stream sat idle for the 0.2 ms the host needed for them, with 0.7 ms of its own chain still to come)
        // (measured and NOT kept, r06, profiles/r06_train_fanout_ab.txt: the blocks' parameter-gradient tails -- a quarter of each
        //  block, needed by nobody downstream -- on a third stream and the image pool's backward on a fourth, started when the image
        //  block's dproxy is final: 1.66 -> 1.87 ms.  Kernels of four streams side by side stretch far beyond fair sharing on this
        //  stack, as the lanes of the eval path did)
        PTX_TRY(record_wait(ev_fork, main, side));
        PTX_TRY(ptx_train_block_bwd(&ib, side));
        PTX_HIP(hipEventRecord(ev_pp, side));
        PTX_TRY(ptx_train_block_bwd(&tb, main));
        PTX_TRY(ptx_train_imgpool_bwd(&ip, side));
        PTX_HIP(hipStreamWaitEvent(main, ev_pp, 0));
        dpp = at<float>(D, L.dpp);
        PTX_TRY(ptx_op_eltwise(0, tb.dx, ib.dx, 0.0f, R * C, 1, dpp, main));
    } else {
        PTX_TRY(ptx_train_block_bwd(&ib, main));
        if (side) PTX_TRY(record_wait(ev_fork, main, side));
        PTX_TRY(ptx_train_imgpool_bwd(&ip, side ? side : main));
        tb.dx_add = ib.dx;                                   // dx = both blocks' gradients of the point proxies
        PTX_TRY(ptx_train_block_bwd(&tb, main));
        dpp = tb.dx;
    }
    // ---- point encoder, kept centres, offset network
    float *dbg_enc = gptr(a, L, PTX_TS_ENC_BN_B);           // (2,W): [dbeta | dgamma]
    PTX_TRY(ptx_op_slotnet_bwd(at<float>(A, L.kcenter), at<float>(A, L.kcluster), R, K, a.enc.W, a.enc.conv_w, a.enc.conv_b, a.enc.bn_w,
                               a.enc.bn_b, at<float>(A, L.mr_enc), 1, at<int32_t>(A, L.arg_enc), dpp, gptr(a, L, PTX_TS_ENC_CONV_W),
                               gptr(a, L, PTX_TS_ENC_CONV_B), dbg_enc, at<float>(D, L.dkc_enc), at<void>(D, L.sc_bwd), L.sc_enc_bytes, main));
    float *dkcenter = at<float>(D, L.dkcenter);
    PTX_TRY(ptx_op_eltwise(0, at<float>(D, L.dkc_enc), dc, 0.0f, R * 3, 1, dkcenter, main));
    float *dcenters = at<float>(D, L.dcenters);
    PTX_HIP(hipMemsetAsync(dcenters, 0, (size_t)BM * 3 * 4, main));          // rows that were not kept get 0
    PTX_TRY(ptx_op_rows_scatter(dkcenter, at<int32_t>(A, L.src), R, 3, dcenters, main));
    PTX_TRY(ptx_op_head_bwd(dcenters, at<float>(A, L.dcoef), at<float>(A, L.pooled), a.map_w, (int)BM, a.off.W, 3, at<float>(D, L.dpooled),
                            gptr(a, L, PTX_TS_MAP_W), at<float>(D, L.hb_tmp), L.hb_tmp_f, main));
    PTX_TRY(ptx_op_slotnet_bwd(at<float>(A, L.c0), at<float>(A, L.cl1), BM, K, a.off.W, a.off.conv_w, a.off.conv_b, a.off.bn_w, a.off.bn_b,
                               at<float>(A, L.mr_off), 0, nullptr, at<float>(D, L.dpooled), gptr(a, L, PTX_TS_OFF_CONV_W),
                               gptr(a, L, PTX_TS_OFF_CONV_B), gptr(a, L, PTX_TS_OFF_BN_B), nullptr, at<void>(D, L.sc_bwd), L.sc_off_bytes, main));
    if (side) PTX_TRY(record_wait(ev_join, side, main));
    return PTX_OK;
}

}  // extern "C"
