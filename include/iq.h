/*
 * iq.h - C ABI of libiq_hip.so, the MI355X (gfx950) implementation of the Shapley-value /
 * multi-order-interaction hot path of ada-shen/Interpret_quality.
 *
 * The reference has no FFI of its own: the path sits behind Python functions.
 * Each entry point below therefore names the reference function (file:line, relative to the
 * reference root) whose device work it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()),
 *    row-major contiguous, unless the parameter is documented as "host";
 *  - no allocation inside the library: scratch memory is a caller-provided workspace;
 *  - every launch function takes a hipStream_t (passed as void*; NULL = default stream) and is
 *    asynchronous with respect to the host;
 *  - return value: IQ_OK (0) or a negative IQ_E* code; a message for the calling thread's last
 *    error is available from iq_last_error();
 *  - indices are int32 at the ABI (the Python wrappers convert from int64), coalitions are
 *    uint64 bit masks over regions (bit r set = region r kept), so num_regions <= 64.
 */
#ifndef IQ_H_
#define IQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* iq_stream_t; /* hipStream_t */

enum {
    IQ_OK = 0,
    IQ_EINVAL = -1,      /* bad argument (shape, null pointer, unsupported size) */
    IQ_ELAUNCH = -2,     /* HIP launch or runtime error */
    IQ_EWORKSPACE = -3,  /* workspace too small */
    IQ_EUNSUPPORTED = -4
};

#define IQ_MAX_REGIONS 64
#define IQ_MAX_POINTS 4096
#define IQ_NUM_FEAT 1024

/* ABI version: 100 * major + minor.  101 (round 4): every weight descriptor (iq_dense_layer, iq_pointnet_weights, ...) gained
 * optional `*_bf3` fields (weights split into three bf16 terms, iq_pack_weight_bf3).  Descriptors MUST be zero-initialised before
 * they are filled (`iq_dense_layer l = {0};` / memset): a NULL *_bf3 pointer selects the float32-MFMA kernels, a non-NULL one is
 * dereferenced on the device.  A client built against an older header must be rebuilt (the structs grew).
 * 102 (round 5): no layout change; a bf16x3 weight image (iq_pack_weight_bf3) now has its k range padded to a multiple of 32 (only images of layers with cin % 32 == 16 differ, and no kernel consumed those before), dense
 * layers take the image for any cin >= 32 and cout = 256 n or 256 n + 64, iq_knn uses a larger tmp when it is given one, PointNet
 * takes clouds of up to IQ_MAX_POINTS points and PointConv of 64 and more. */
#define IQ_ABI_VERSION 102
int iq_version(void);
const char* iq_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Coalition masking
 * ------------------------------------------------------------------------------------------- */

/* Shapley prefix masking: replaces tools/final_common.py:46-61 (mask_data_batch) together with
 * the expand().clone() at :88, and final_shapley_value.py:74-88 (mask_data) for bs = 1.
 * Row o*(R+1)+i of `out` keeps the regions orders[o][0..i-1]; every other point := center.
 * out is (bs*(R+1), N, 3) if channel_first == 0, else (bs*(R+1), 3, N)
 * (the layout after the permute().contiguous() of tools/final_common.py:35). */
int iq_mask_shapley(const float* cloud /*N,3*/, const int32_t* region_id /*N*/,
                    const int32_t* orders /*bs,R*/, const float* center /*3*/, float* out,
                    int N, int R, int bs, int channel_first, iq_stream_t stream);

/* Interaction masking: replaces final_point_binary_interaction_logits.py:45-56.
 * For context k: rows 4k..4k+3 of out (4*nb, 3, N) keep S+{i,j}, S+{i}, S+{j}, S where
 * S = ctx_mask[k] (bit mask over regions) and (i, j) = pairs[k]; masked entries := center. */
int iq_mask_interaction(const float* cloud /*N,3*/, const int32_t* region_id /*N*/,
                        const int32_t* pairs /*nb,2*/, const uint64_t* ctx_mask /*nb*/,
                        const float* center /*3*/, float* out /*4nb,3,N*/,
                        int N, int R, int nb, iq_stream_t stream);

/* Generic coalition masking from explicit keep masks: out row b keeps the regions in keep[b].
 * (Used by the models that consume materialised clouds; PointNet never materialises them.) */
int iq_mask_coalitions(const float* cloud /*N,3*/, const int32_t* region_id /*N*/,
                       const uint64_t* keep /*B*/, const float* center /*3*/, float* out,
                       int N, int B, int channel_first, iq_stream_t stream);

/* Validation of index inputs (region ids in [0,R), permutation / pair entries in [0,R), FPS indices in [0,N)):
 * returns IQ_EINVAL and names the first offending position if any idx[i] lies outside [lo, hi).  The ONE entry point
 * that synchronises `stream` (it reads a 4-byte verdict back); `scratch` = 4 bytes of device memory.  The launch
 * functions themselves never fault on an out-of-range id - such a point is treated as belonging to no region (always
 * masked), an out-of-range order entry is ignored - but their results are then meaningless; callers that take ids from
 * files (region_id.npy, all_orders.npy: tools/final_common.py:120-121) validate once per cloud with this call. */
int iq_check_index_range(const int32_t* idx, size_t count, int lo, int hi, uint32_t* scratch /*4 B, device*/,
                         iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Coalition sampling (on the device)
 * ------------------------------------------------------------------------------------------- */

/* final_shapley_value.py:59-72 (generate_all_orders): S permutations of 0..R-1 drawn from NumPy's legacy global
 * generator, i.e. MT19937 + RandomState.shuffle's back-to-front Fisher-Yates with masked rejection sampling, CONTINUED
 * ON THE DEVICE: mt_state is the generator's state as np.random.get_state() returns it - 624 key words followed by the
 * position (625 uint32, device memory) - and is advanced in place, so permutations and every later host draw (after
 * np.random.set_state with the returned words) are bit-identical to the reference's stream for the same seed.
 * One workgroup (the stream is sequential); ~0.2 ms for 1000 permutations of 32 regions. */
int iq_sample_permutations(uint32_t* mt_state /*625*/, int32_t* orders /*S,R*/, int S, int R, iq_stream_t stream);

/* tools/final_common.py:56-60 as bit masks: keep[s*(R+1) + i] = the regions orders[s][0..i-1]
 * (row 0 = nothing kept = all-centre cloud, row R = the unmodified cloud). */
int iq_prefix_keep_masks(const int32_t* orders /*S,R*/, uint64_t* keep /*S*(R+1)*/, int S, int R, iq_stream_t stream);

/* final_point_binary_interaction_logits.py:45-52 as bit masks: for pair p = (i, j) and its context c = contexts[p][c][0..m-1]
 * (np.in1d(region_id, context)): keep[4(pC + c) + 0..3] = S+{i,j}, S+{i}, S+{j}, S.  m = 0: the empty context. */
int iq_context_keep_masks(const int32_t* pairs /*P,2*/, const int32_t* contexts /*P,C,m*/, uint64_t* keep /*4PC*/,
                          int P, int C, int m, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Reward and reductions
 * ------------------------------------------------------------------------------------------- */

/* tools/final_common.py:11-24 (get_reward).  modified != 0: v = z_y - logsumexp(z_{!=y});
 * modified == 0 ("normal"): v = log_softmax(z)[y]. */
int iq_reward(const float* logits /*B,C*/, int label, int modified, float* v /*B*/,
              int B, int C, iq_stream_t stream);

/* tools/final_common.py:92-96 and final_shapley_value.py:145-150: dv = v[1:] - v[:-1] per
 * permutation (float32), scattered by the permutation into float64 rows and summed over the
 * permutations IN ORDER (bit-stable, matches the host loop of the reference).
 *   sv_rows  (S,R) float64  = the rows of region_sv_all.npy        (may be NULL)
 *   phi_sum  (R,)  float64  = sum over the S permutations (not divided)
 *   snaps    (n_snap,R) float64 = running sums after snap_counts[k] permutations (may be NULL),
 *            i.e. what save_shapley (final_shapley_value.py:91-106) divides by count. */
int iq_shapley_accum(const float* v /*S*(R+1)*/, const int32_t* orders /*S,R*/,
                     double* sv_rows, double* phi_sum, const int32_t* snap_counts, int n_snap,
                     double* snaps, int R, int S, iq_stream_t stream);

/* final_cal_interactions.py:28-36: I[k] = ((v[4k] + v[4k+3]) - v[4k+1]) - v[4k+2] in float32. */
int iq_interaction_reduce(const float* v /*4n*/, float* out /*n*/, int n, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Geometry
 * ------------------------------------------------------------------------------------------- */

/* final_shapley_value.py:20-35 (cal_region_id) with tools/final_util.py:134-147: nearest FPS
 * centre per point by the expanded-form squared distance, first index on ties. */
int iq_region_assign(const float* cloud /*N,3*/, const int32_t* fps_idx /*R*/,
                     int32_t* region_id /*N*/, int N, int R, iq_stream_t stream);

/* final_save_fps.py:10-31 (= models/pointnet2.py:45-68): farthest point sampling, start index 0,
 * ties -> lowest index.  xyz (B,N,3) -> idx (B,S). */
int iq_fps(const float* xyz, int32_t* idx, int B, int N, int S, iq_stream_t stream);

/* final_smoothness_center_enum_all.py:183-242 (update_region) for every region and every epoch of
 * test_all_region's loop (:303-335) in one launch: gradient ascent (objective +1) / descent (-1) on the
 * linearity (mode 0) / planarity (1) / scattering (2) of each region's points, under the reference's
 * variance bounds (:66-82), distance bound (:102-118) and stop conditions (:163-180).  The constants are
 * the reference's module-level STEP .. MAX_ITERATION (:13-19).  Regions only move their own points, so
 * epoch e of region r does not depend on other regions.  Outputs:
 *   data_out   (epochs,N,3)  the cloud after epoch e (data_list, :326)
 *   smooth_out (epochs,R)    smoothness of region r recorded in epoch e (smoothness_list, :325)
 *   var_out    (epochs,R,3)  last variances on o1,o2,o3 (the "var1: .." log line, :239)
 *   orig_out   (R,4)         original variances on o1,o2,o3 and original smoothness (:256-261)
 *   stop_epoch (R)           epoch in which the region stopped updating (indicator False), `epochs` if it
 *                            never did, -1 for a region of fewer than two points (left untouched;
 *                            the reference cannot process such a region)
 * The caller evaluates epochs 0 .. min(epochs, max_r stop_epoch + 1) - 1, as the break at :333-334 does.
 * `origin` (optional) restarts the enumeration from an already deformed `cloud`: orientations, variance
 * bounds, the distance bound and the first target are taken from `origin` (NULL: from `cloud`). */
typedef struct iq_smoothness_params {
    double step;           /* STEP = 1e-3 */
    double enum_step;      /* ENUM_STEP = 0.05 */
    double var_threshold;  /* VAR_THRESHOLD = 0.003 */
    double dist_threshold; /* DIST_THRESHOLD = 0.03 */
    double stop_ratio;     /* STOP_RATIO = 0.5 */
    int32_t epochs;        /* EPOCH = 50 */
    int32_t max_iteration; /* MAX_ITERATION = 100 */
    int32_t project_to_bound; /* 0 = as the reference behaves: points beyond dist_threshold are counted (stop
                                 condition) but stay where they are, because its write-back at :117 assigns to a
                                 temporary view; 1 = project them back onto the sphere as :110-117 intends */
    int32_t reserved;
} iq_smoothness_params;
int iq_smoothness_enum(const float* cloud /*N,3*/, const float* origin /*N,3 or NULL*/, const int32_t* region_id /*N*/, int N, int R, int mode,
                       int objective, const iq_smoothness_params* prm, float* data_out, float* smooth_out,
                       float* var_out, float* orig_out, int32_t* stop_epoch, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PointNet (models/pointnet.py:11-115) - fused fp32-MFMA forward over coalitions
 * ------------------------------------------------------------------------------------------- */

/* One dense layer y = act(W x + b), BatchNorm(eval) already folded in.  `w` is in the library's
 * MFMA B-fragment order (iq_pack_weight), `b` has cout_padded entries. */
typedef struct iq_dense_layer {
    const float* w;
    const float* b;
    int32_t cin;
    int32_t cout;
    const void* w_bf3;   /* optional (NULL: fp32 MFMA): the same weights as three bf16 terms (iq_pack_weight_bf3); wide layers
                          * (cout a multiple of 256, cin >= 32) then take their products on the bf16 matrix pipe,
                          * float32-exact; cout = 256 n + 64: the first 256 n columns do, the last 64 stay on the fp32 MFMA */
} iq_dense_layer;

typedef struct iq_pointnet_weights {
    const float* stn_in;  /* [64][4] = (w0,w1,w2,bias) of feat.stn.conv1+bn1 */
    iq_dense_layer stn_c2, stn_c3, stn_fc1, stn_fc2, stn_fc3;       /* fc3 bias includes +I */
    const float* feat_in; /* [64][4] of feat.conv1+bn1 */
    iq_dense_layer fstn_c1, fstn_c2, fstn_c3, fstn_fc1, fstn_fc2, fstn_fc3; /* fc3: iq_pack_fstn_fc3.  feature_transform = False
        (models/pointnet.py:62-63): fstn_c1.w = NULL and fstn_fc3.b = the packed identity (iq_pack_fstn_fc3 of a zero layer) */
    iq_dense_layer feat_c2, feat_c3, cls_fc1, cls_fc2, cls_fc3;
    /* optional (NULL: fp32 MFMA): the folded 128 -> 1024 layers of the feature STN and of the trunk as three bf16 terms
     * (iq_pack_weight_bf3): layer 3 of the coalition chains - 91 % of their work - then runs on the bf16 matrix pipe, float32-exact */
    const void* fstn_c3_bf3;
    const void* feat_c3_bf3;
    /* optional, used together with the two above: the 64 -> 128 layers before them, likewise */
    const void* fstn_c2_bf3;
    const void* feat_c2_bf3;
} iq_pointnet_weights;

/* out (M,ldo) = act(A (M,lda) . W^T + b): the dense layer every model kernel shares (1x1 convolutions and
 * fully connected layers with BatchNorm folded: models/pointnet.py:24-29, models/dgcnn.py:66-80,
 * models/pointnet2.py:163-170 ...), exposed for tests and callers that fold their own layers.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2).  Device pointers; L->cin % 8 == 0. */
int iq_linear(const float* A, int lda, const iq_dense_layer* L, float* out, int ldo, int M, int act,
              iq_stream_t stream);

/* Host-side packing helpers (host pointers).  iq_packed_floats = number of floats of the packed
 * image of a (cout, cin) weight; cin must be a multiple of 8. */
size_t iq_packed_floats(int cout, int cin);
int iq_padded_cout(int cout);
int iq_pack_weight(const float* w_host /*cout,cin*/, float* out_host, int cout, int cin);
/* The same weight as THREE bf16 terms (h = bf16(w), m = bf16(w - h), l = bf16(w - h - m), round to nearest even: 24 mantissa bits)
 * in the fragment order of v_mfma_f32_32x32x16_bf16: [term][n-tile][k-step of 16][lane][8], iq_packed_bf3_elems 16-bit elements.
 * A layer that is given it (iq_dgcnn_weights.conv5_bf3) takes its products on the bf16 matrix pipe - six exact bf16 products
 * per float32 product, float32 accumulation: float32 accuracy at 3/8 of the matrix cycles of the fp32 MFMA.  cin % 8 == 0; the image's
 * k range is cin rounded up to a multiple of 32, zero beyond cin (the dense layer never reads A's columns there). */
size_t iq_packed_bf3_elems(int cout, int cin);
int iq_pack_weight_bf3(const float* w_host /*cout,cin*/, unsigned short* out_host, int cout, int cin);
/* feat.fstn.fc3 (4096 x 256): permutes the output rows so that the layer's output vector IS the
 * packed B-fragment image of trans_feat for the trunk's per-coalition 64x64 product, and adds the
 * identity (models/pointnet.py:42-45) into the bias.  out_w has iq_packed_floats(4096,256)
 * floats, out_b 4096.  perm_host[r] (4096 int32, may be NULL) receives the source row of output
 * row r, so a caller can undo the permutation. */
int iq_pack_fstn_fc3(const float* w_host, const float* b_host, float* out_w, float* out_b,
                     int32_t* perm_host);

/* Workspace (bytes) needed by iq_pointnet_coalitions for B coalitions over nclouds clouds. */
size_t iq_pointnet_workspace_bytes(int B, int nclouds, int N, int R);

/* Logits of B coalitions without materialising the masked clouds.  Replaces, for PointNet, the
 * chain mask_data_batch -> cal_reward -> model(...) of tools/final_common.py:88-91 and
 * final_point_binary_interaction_logits.py:45-60.
 *   clouds    (nclouds, N, 3) if channel_first == 0, else (nclouds, 3, N)
 *   centers   (nclouds, 3)   the point every masked point collapses onto
 *   region_id (nclouds, N)   int32 in [0, R)
 *   keep      (B,) uint64    bit r set = region r keeps its points
 *   cloud_of  (B,) int32     cloud index of each coalition (NULL = all 0)
 *   logits    (B, 10)
 *   trans_feat_packed (B, 4096) optional output (NULL to skip): packed image of trans_feat
 * Exact with respect to the dense evaluation of the same kernels: identical points produce
 * identical features and max-pooling ignores duplicates. */
int iq_pointnet_coalitions(const iq_pointnet_weights* w /*host struct of device pointers*/,
                           const float* clouds, const float* centers, const int32_t* region_id,
                           const uint64_t* keep, const int32_t* cloud_of, float* logits,
                           float* trans_feat_packed, void* workspace, size_t workspace_bytes,
                           int B, int nclouds, int N, int R, int channel_first,
                           iq_stream_t stream);

/* The same with PointNetfeat's `crt_points` (models/pointnet.py:83: the arg-max of the trunk's max-pool, third element of
 * PointNetCls's output tuple): crt_points (B, 1024) int32 receives, per coalition and channel, the index of the point that
 * attains the maximum (largest value, then the first in region-sorted row order = lowest point index for the dense
 * forward, where all points share one region); index N stands for the centre that masked points collapse to.  NULL = the
 * entry point above.  The hot-path callers discard crt_points (tools/final_common.py:36-37); the arg-max variant of the
 * trunk kernel costs a few percent and is only instantiated for this entry point. */
int iq_pointnet_coalitions_crt(const iq_pointnet_weights* w, const float* clouds, const float* centers,
                               const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                               float* trans_feat_packed, int32_t* crt_points, void* workspace, size_t workspace_bytes,
                               int B, int nclouds, int N, int R, int channel_first, iq_stream_t stream);

/* Algorithmic FLOP of the dense reference network per coalition (0.879 GFLOP at
 * N = 1024), the basis of bench.py's roofline figures. */
double iq_pointnet_flops_per_coalition(int N);

/* ---------------------------------------------------------------------------------------------
 * PointNet++ MSG (models/pointnet2.py:12-276)
 * ------------------------------------------------------------------------------------------- */

/* models/pointnet2.py:70-91 (query_ball_point): for every centroid the K LOWEST-INDEX points with
 * squared distance <= radius^2 (expanded form -2 c.p + |c|^2 + |p|^2), padded with the first hit.
 * xyz (B,N,3), new_xyz (B,S,3) -> idx (B,S,K) int32. */
int iq_ball_query(const float* xyz, const float* new_xyz, float radius, int K, int32_t* idx,
                  int B, int N, int S, iq_stream_t stream);

/* One scale of a multi-scale set abstraction: layer 1 = relu(w1x . (x_p - c) [+ U_p]), then the
 * dense layers l2, l3 (BN folded, packed), max over the `nsample` members. */
typedef struct iq_pn2_scale {
    const float* w1x;      /* [C1][4] = (wx0, wx1, wx2, bias); bias = 0 when the per-point part U carries it */
    iq_dense_layer l2, l3;
    float radius;
    int32_t nsample;
} iq_pn2_scale;

typedef struct iq_pointnet2_weights {
    iq_pn2_scale sa1[3];
    iq_dense_layer sa2_u;  /* 320 -> 64+128+128: feature part of layer 1 of the three sa2 scales (+bias) */
    iq_pn2_scale sa2[3];
    iq_dense_layer sa3_l1, sa3_l2, sa3_l3;  /* 643 (zero-padded to 648: [xyz, features]) -> 256 -> 512 -> 1024 */
    iq_dense_layer fc1, fc2, fc3;
    /* optional (NULL: fp32 MFMA): layers 2 and 3 of an sa2 scale as three bf16 terms (iq_pack_weight_bf3); a 128-128-256 scale
     * that has both runs its grouped MLP on the bf16 matrix pipe, float32-exact (pn2_group_bf3_kernel) */
    const void* sa2_l2_bf3[3];
    const void* sa2_l3_bf3[3];
} iq_pointnet2_weights;

size_t iq_pointnet2_workspace_bytes(int B);

/* Eval-mode PointNet2ClsMsg.forward (models/pointnet2.py:264-276) on B materialised clouds.
 * xyz (B,N,3) channel-LAST (what iq_mask_* writes with channel_first = 0) -> logits (B,10). */
int iq_pointnet2_forward(const iq_pointnet2_weights* w /*host struct of device pointers*/, const float* xyz,
                         float* logits, void* workspace, size_t workspace_bytes, int B, int N,
                         iq_stream_t stream);

/* The same network on B coalitions given as region bit masks (argument convention of iq_pointnet_coalitions):
 * replaces the masking + forward of tools/final_common.py:88-91 and
 * final_point_binary_interaction_logits.py:45-60 for PointNet++.  The masked clouds are written internally
 * (FPS, ball query and sa2 / sa3 run on them as in iq_pointnet2_forward), but sa1 - which has no input
 * features, so a member row is a function of the point pair only - is a gather-max over per-source-cloud tables
 * MLP_s(P[q] - P[p]) of all pairs inside each radius (P = the cloud's points plus the centre masked points
 * collapse to), built once per call with the ball query's own distance expression and the grouped kernel's layer
 * arithmetic: bit-identical features.  A scale whose pair count exceeds the table capacity (192 pairs per point
 * on average) falls back to the grouped MLP.  Reads the pair counts back once per call (the only host round
 * trip); otherwise asynchronous on `stream`. */
size_t iq_pointnet2_coalitions_workspace_bytes(int B, int nclouds, int N);
int iq_pointnet2_coalitions(const iq_pointnet2_weights* w, const float* clouds, const float* centers,
                            const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of,
                            float* logits, void* workspace, size_t workspace_bytes, int B, int nclouds, int N,
                            iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DGCNN / GCNN (models/dgcnn.py:12-194)
 * ------------------------------------------------------------------------------------------- */

/* models/dgcnn.py:12-18 (knn): the k = 20 largest of -|x_i|^2 - (-2 x_i.x_j) - |x_j|^2 per row, self
 * included, as an (unordered) index set.  x (B,N,C) row-major with C in {3, 64, 128}; idx (B,N,20)
 * int32; tmp = scratch of at least B*N*84 + 16*B + 8192 bytes; with B*N*C*6 bytes more (C = 64, 128) the inner products run on
 * the bf16 matrix pipe as three-term bf16 products, float32-accurate, as they do inside iq_dgcnn_* (csrc/iq_dgcnn.hip: knn_kernel
 * <.., BF3>; iq_set_tuning(5, 22): the fp32 MFMA kernels).  Both arithmetics stand in front of the same exact re-ranking:
 * Feature-space graphs (C = 64, 128): where the float32 expanded form cannot separate the 20th from the 21st nearest
 * (gap below 1e-5 of the magnitude of the summed terms: rounding noise decides for the reference's float32 path too), the
 * query's 21 best candidates are re-ranked by -sum (x_i - x_j)^2 accumulated in float64, i.e. the order the reference
 * finds in float64 (csrc/iq_dgcnn.hip: knn_refine_kernel, 0.7 % of the queries; iq_set_tuning(5, 20) keeps the float32
 * ranking, for A/B runs).  Standing in front of that re-ranking, the float32 selection of these two cases compares distances in
 * buckets of 32 ulps (3.8e-6 relative; csrc/iq_topk.h: TaggedTopK) - a fifth of the narrowest re-ranked band; C = 3 is exact. */
int iq_knn(const float* x, int32_t* idx, void* tmp, size_t tmp_bytes, int B, int N, int C, int k,
           iq_stream_t stream);

/* pq[l]: layer l's EdgeConv as ONE dense layer over the points with 2*Cout outputs
 * [P = (s.W_a) x ; Q = (s.(W_b - W_a)) x + t] (BN folded; see csrc/iq_dgcnn.hip); conv5/fc1/fc2 are
 * followed by LeakyReLU(0.2). */
typedef struct iq_dgcnn_weights {
    iq_dense_layer pq[4];
    iq_dense_layer conv5, fc1, fc2, fc3;
    int32_t k;
    int32_t reserved;
    const void* conv5_bf3;    /* optional (NULL: fp32 MFMA): conv5's folded weights as three bf16 terms, iq_pack_weight_bf3 */
} iq_dgcnn_weights;

size_t iq_dgcnn_workspace_bytes(int B, int N);

/* Eval-mode DGCNN_cls.forward (models/dgcnn.py:83-120; fixed_graph = 0) or GCNN_cls.forward
 * (:156-194; fixed_graph = 1: one xyz graph for all four EdgeConv layers) on B materialised clouds.
 * xyz (B,N,3) channel-last -> logits (B,10).  Any 20 <= N <= 32767 (clouds are padded to a multiple of 32 rows
 * internally with rows no query can select). */
int iq_dgcnn_forward(const iq_dgcnn_weights* w /*host struct of device pointers*/, const float* xyz,
                     float* logits, void* workspace, size_t workspace_bytes, int B, int N,
                     int fixed_graph, iq_stream_t stream);

/* The same network on B coalitions WITHOUT materialising the masked clouds (the argument convention of
 * iq_pointnet_coalitions: clouds (nclouds,N,3), centers (nclouds,3), region_id (nclouds,N), keep (B) region
 * bit masks, cloud_of (B) or NULL when nclouds is 1 or B).  Replaces the masking + forward of
 * tools/final_common.py:88-91 and final_point_binary_interaction_logits.py:45-60 for DGCNN / GCNN.  A masked
 * cloud is its kept points plus M copies of the centre that behave identically in every layer, so each
 * coalition runs on kept + min(M,20) rows (exact: neighbourhood max and max-pool are set operations, the
 * mean pool weights the centre by M).  Same workspace as iq_dgcnn_forward(B, N). */
int iq_dgcnn_coalitions(const iq_dgcnn_weights* w, const float* clouds, const float* centers,
                        const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                        void* workspace, size_t workspace_bytes, int B, int nclouds, int N, int fixed_graph,
                        iq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PointConv with density (models/pointconv.py:103-424)
 * ------------------------------------------------------------------------------------------- */

/* One PointConvDensitySetAbstraction (BN folded).  Layer 1 of the shared MLP is split:
 * relu(w1x . (x_p - c) + U_p) with U = u(features) (bias inside; sa1 has no features: bias in w1x[3]).
 * densitynet: rows [w | b] of 1->16->8->1 (32 + 136 + 9 floats); weightnet: 3->8->8->16 (32 + 72 + 144). */
typedef struct iq_pointconv_sa {
    const float* w1x;         /* [C1][4] = (wx0, wx1, wx2, bias) */
    iq_dense_layer u;         /* features -> C1 (unused for sa1) */
    iq_dense_layer l2, l3;
    const float* densitynet;
    const float* weightnet;
    iq_dense_layer linear;    /* 16*C3 -> C3, + bn_linear, ReLU */
    float bandwidth;
    int32_t nsample;          /* 32, 64, 0 = group all */
} iq_pointconv_sa;

typedef struct iq_pointconv_weights {
    iq_pointconv_sa sa[3];
    iq_dense_layer fc1, fc2, fc3;
    /* optional (NULL: fp32 MFMA): layers 2 and 3 of sa[1] (128 -> 128 -> 256) as three bf16 terms (iq_pack_weight_bf3): its
     * grouped MLP then runs on the bf16 matrix pipe, float32-exact (pc_group_bf3_kernel) */
    const void* sa2_l2_bf3;
    const void* sa2_l3_bf3;
} iq_pointconv_weights;

size_t iq_pointconv_workspace_bytes(int B, int N);

/* Eval-mode PointConvDensityClsSsg.forward (models/pointconv.py:414-424) on B materialised clouds.
 * xyz (B,N,3) channel-last -> logits (B,10). */
int iq_pointconv_forward(const iq_pointconv_weights* w /*host struct of device pointers*/, const float* xyz,
                         float* logits, void* workspace, size_t workspace_bytes, int B, int N,
                         iq_stream_t stream);

/* Logits of B coalitions given as region bit masks over nclouds source clouds - the same call as
 * iq_pointnet2_coalitions / iq_dgcnn_coalitions; replaces mask_data_batch + model(...) of tools/final_common.py:46-61,
 * 26-43 and final_point_binary_interaction_logits.py:45-60 for PointConv.  The masked clouds are written into the
 * workspace and run through the forward of iq_pointconv_forward.  When a few source clouds serve many coalitions
 * (nclouds <= 8, or nclouds * 8 <= B) the K-nearest groups of sa1 and sa2 (models/pointconv.py:103-114) are read off per-source-cloud
 * sorted neighbour lists, built once per call with the kNN kernel's own distance expression: in xyz space neither the
 * distance between two points nor the centre depends on the coalition, only the candidate set does (the same point
 * sets as the kNN kernel up to ties; masked points are interchangeable).  With at most eight source clouds sa1's MLP rows
 * - functions of (member point, centroid point) only, sa1 having no input features - come from a table of all (N+1)^2
 * pairs (0.54 GB of workspace per source cloud) and a group is 32 table rows contracted with the members' density x
 * WeightNet weights, in one kernel with the 2048 -> 128 layer behind it.  64 <= N <= 1024 (fewer than 512 points: sa1's
 * farthest point sampling returns index 0 once the points are used up, models/pointconv.py:54-77, and so does this).  Asynchronous on `stream`. */
size_t iq_pointconv_coalitions_workspace_bytes(int B, int nclouds, int N);
int iq_pointconv_coalitions(const iq_pointconv_weights* w, const float* clouds, const float* centers,
                            const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                            void* workspace, size_t workspace_bytes, int B, int nclouds, int N, iq_stream_t stream);
/* The same call for several launches on the SAME source clouds (the chunks of one interaction setting,
 * final_point_binary_interaction_logits.py:37-66; the batches of one pose, tools/final_common.py:78-96).  The per-cloud
 * structures (padded rows, sorted neighbour lists, pair tables) occupy the first iq_pointconv_tables_bytes(nclouds, N) bytes of
 * `workspace`, whatever B is.  `tables_state` (HOST int, in / out; NULL = build everything, keep nothing): bit 0 = the lists,
 * bit 1 = the pair tables are already there for exactly these clouds, centers, nclouds and N - the caller's promise (same
 * workspace base, bytes untouched since the call that set the bit).  Missing parts are built and their bits set.
 * Bits 2-3 (in, returned unchanged) name how the K-nearest groups are formed: 0 = decided from THIS launch (the sorted-list walk
 * when nclouds <= 8 or nclouds * 8 <= B, else a per-coalition kNN), 1 = always the walk, 2 = never.  The two sum a group's members
 * in a different order (results equal to rounding, not bitwise): a caller that splits ONE batch over several launches and wants
 * launch-size-independent bits passes 1 or 2 on every launch (interpret_quality_amd/pointconv.py decides from the whole batch). */
size_t iq_pointconv_tables_bytes(int nclouds, int N);
int iq_pointconv_coalitions_cached(const iq_pointconv_weights* w, const float* clouds, const float* centers,
                                   const int32_t* region_id, const uint64_t* keep, const int32_t* cloud_of, float* logits,
                                   void* workspace, size_t workspace_bytes, int B, int nclouds, int N, int* tables_state,
                                   iq_stream_t stream);

/* The diagnostic entry points (HIP-event profiler, experiment knobs, debug counters) are NOT part of the drop-in surface:
 * they are declared in iq_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* IQ_H_ */
