// Training-label rasterisation on the device: the per-COLUMN half of reference dataset.py:108-120,137-169 (cor_2_1d, the 1-D wall-wall
// corner probability) and misc/panostretch.py:51-78 (pano_connect_points).
//
// Split (horizonnet_amd/labels.py, device_label_records): the per-CORNER scalars of every wall edge -- longitude / latitude, the
// ray length z / tan v, the edge's end point and direction on the z plane, all in the reference's float32 scalar flow -- are
// <= ~30 x 8 scalar numpy operations per panorama and stay on the host, bit for bit what the reference computes.  Everything that
// is per column (1024 x edges: tan of the column's longitude, the ray-edge intersection, sqrt, arctan2, the one-row-per-column
// selection, flip / roll of the finished vectors, 0.96 ** distance) runs here in float64 in the reference's operation order, FMA
// contraction off.  That moves ~1.5 ms of host time per panorama (at 8 ranks per host: most of a core per rank) to ~10 us of GPU.
//
// Selection of one row per column (dataset.py:137-147): the reference sorts all trace points by x + y / y.max() * (+1 ceiling | -1
// floor) and keeps the first of every x; here every column takes the minimum of the same key over the edges that cover it (ties
// cannot be told apart in the reference either: its argsort is not stable).  np.interp over the kept points is the identity at
// every integer column that has a point; a column NO edge covers (an open outline) sets status[b] = 1 and the caller rasterises
// that panorama on the host instead.
//
// Record per panorama (floats): [0] edges ceiling, [1] edges floor, [2] visible corners, [3] flip, [4] roll, [5..7] 0;
// then edge[2][max_seg][8] = {kind, x1, y1, dx, dy, first column, columns, z} (kind 1: both corners on one column: x1 = that column
// (may be fractional), y1 / dx = the two rows); then corner_x[max_cor] (after flip / roll, as dataset.py:88-97 leaves them).
#include "hn_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int LB_HDR = 8;
constexpr int LB_SEG = 8;

__global__ __launch_bounds__(256) void labels_rasterise_kernel(const float* __restrict__ rec, int rec_floats, int max_seg, int max_cor, int H, int W,
                                                               double p_base, float* __restrict__ bon, float* __restrict__ y_cor, int* __restrict__ status)
{
    extern __shared__ float seg_s[];                        // this boundary's edges
    __shared__ double red[256];
    const int b = blockIdx.y;
    const int bd = blockIdx.x;                              // 0 ceiling, 1 floor, 2 corner probability
    const float* r = rec + (size_t)b * rec_floats;
    const int flip = (int)r[3], roll = (int)r[4];
    const int tid = threadIdx.x;

    if (bd == 2) {
        const int nc = (int)r[2];
        const float* cx = r + LB_HDR + 2 * max_seg * LB_SEG;
        for (int j = tid; j < W; j += 256) {
            double dmin = 1e300;
            for (int k = 0; k < nc; ++k) {
                const double c = (double)cx[k];
                const double d0 = fabs(c - (double)j), d1 = fabs(c - (double)(j + W)), d2 = fabs(c - (double)(j - W));
                dmin = fmin(dmin, fmin(fmin(d0, d1), d2));
            }
            y_cor[(size_t)b * W + j] = nc > 0 ? (float)pow(p_base, dmin) : 0.f;      // (no visible corner: the caller never sends one; defined anyway)
        }
        return;
    }

    const int ns = (int)r[bd];
    const float* sg = r + LB_HDR + bd * max_seg * LB_SEG;
    for (int i = tid; i < ns * LB_SEG; i += 256) seg_s[i] = sg[i];
    __syncthreads();
    const double sgn = bd == 0 ? 1.0 : -1.0;
    const double pi = 3.141592653589793;

    // row of edge k at integer column c, if the edge covers it (pano_connect_points); kind 1: up to two rows on one column
    auto rows_at = [&](int k, int c, double& ya, double& yb) -> int {
        const float* e = seg_s + k * LB_SEG;
        if (e[0] != 0.f) {
            // a vertical edge sits on ONE column: an integral x, taken modulo W exactly as the host half marks it covered
            // (labels.device_label_record: covered[int(x) % W]); a fractional x touches no integer column
            const double xe = (double)e[1];
            if (xe != floor(xe)) return 0;
            double xw = fmod(xe, (double)W);
            if (xw < 0.0) xw += (double)W;
            if (xw != (double)c) return 0;
            ya = (double)e[2]; yb = (double)e[3];
            return 2;
        }
        const int first = (int)e[5], cnt = (int)e[6];
        int off = (c - first) % W;
        if (off < 0) off += W;
        if (off >= cnt) return 0;
        const double x1 = (double)e[1], y1 = (double)e[2], dx = (double)e[3], dy = (double)e[4], z = (double)e[7];
        const double lon = (((double)c + 0.5) / (double)W - 0.5) * 2.0 * pi;
        const double t = tan(lon);
        const double s = (t * x1 - y1) / (dy - t * dx);
        const double px = x1 + s * dx, py = y1 + s * dy;
        const double rng = sqrt(px * px + py * py);
        ya = (atan2(z, rng) / pi + 0.5) * (double)H - 0.5;
        return 1;
    };

    // pass 1: y.max() over every trace point of this boundary (the fractional-column points of kind-1 edges included)
    double ymax = -1e300;
    for (int c = tid; c < W; c += 256)
        for (int k = 0; k < ns; ++k) {
            double ya, yb;
            const int n = rows_at(k, c, ya, yb);
            if (n >= 1) ymax = fmax(ymax, ya);
            if (n == 2) ymax = fmax(ymax, yb);
        }
    if (tid == 0)
        for (int k = 0; k < ns; ++k)
            if (seg_s[k * LB_SEG] != 0.f) ymax = fmax(ymax, fmax((double)seg_s[k * LB_SEG + 2], (double)seg_s[k * LB_SEG + 3]));
    red[tid] = ymax;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] = fmax(red[tid], red[tid + st]);
        __syncthreads();
    }
    ymax = red[0];

    // pass 2: the first point of every column in the reference's sort order; latitude; flip / roll
    for (int c = tid; c < W; c += 256) {
        double best = 0.0, bkey = 1e300;
        bool found = false;
        for (int k = 0; k < ns; ++k) {
            double ya, yb;
            const int n = rows_at(k, c, ya, yb);
            for (int q = 0; q < n; ++q) {
                const double y = q == 0 ? ya : yb;
                const double key = (double)c + y / ymax * sgn;
                if (!found || key < bkey) { bkey = key; best = y; found = true; }
            }
        }
        if (!found) atomicOr(&status[b], 1);
        const double lat = ((best + 0.5) / (double)H - 0.5) * pi;
        int j = flip ? W - 1 - c : c;
        j = (j + roll) % W;
        bon[((size_t)b * 2 + bd) * W + j] = (float)lat;
    }
}

}  // namespace

// rec: device, B records of rec_floats floats (layout above); bon [B][2][W], y_cor [B][W] float32 (device); status [B] int32 (device):
// 0 = done, 1 = some column of the panorama has no trace point (rasterise it on the host).  One launch per batch.
extern "C" int hn_labels_rasterise(const float* rec, int rec_floats, int max_seg, int max_cor, int B, int H, int W, double p_base,
                                   float* bon, float* y_cor, int* status, void* stream)
{
    if (B == 0) return 0;
    HN_REQUIRE(rec && bon && y_cor && status, "labels_rasterise: null pointer");
    HN_REQUIRE(B > 0 && H > 0 && W > 0 && max_seg > 0 && max_seg <= 256 && max_cor >= 0, "labels_rasterise: bad shape");
    HN_REQUIRE(rec_floats >= LB_HDR + 2 * max_seg * LB_SEG + max_cor, "labels_rasterise: record too short");
    hipStream_t s = (hipStream_t)stream;
    HN_HIP(hipMemsetAsync(status, 0, sizeof(int) * (size_t)B, s));
    hipLaunchKernelGGL(labels_rasterise_kernel, dim3(3, (unsigned)B), dim3(256), sizeof(float) * (size_t)max_seg * LB_SEG, s, rec, rec_floats, max_seg,
                       max_cor, H, W, p_base, bon, y_cor, status);
    HN_LAUNCH_CHECK();
    return 0;
}
