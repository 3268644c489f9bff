// Host-side batched Manhattan fit: the DECISION half of reference misc/post_proc.py gen_ww (gen_ww_general :243-334,
// gen_ww_cuboid :205-240, vote :75-98) and of the validity test + cuboid fallback of inference.py:113-126, for a whole
// batch of panoramas on a thread pool inside the library.  No device code: this is the host half of BASELINE configs[4].
//
// Why native: the per-panorama numpy restatement (horizonnet_amd/postproc.py) costs ~1.3 ms of interpreter time per
// panorama, had to be spread over forked worker processes (fork after HIP initialisation, five pickled arrays per panorama,
// one pool per rank) and caps an 8-GPU node at ~12 k panoramas/s.  Why it is still bit-identical to the reference: every
// TRANSCENDENTAL value (tan / sin / cos / arctan of whole signal rows and of the final corner points) stays in numpy, batched
// over the panoramas by the caller -- numpy's SIMD tan / arctan / arctan2 differ from libm in the last bit, so calling libm
// here would not reproduce the reference -- and this file only performs IEEE +, -, *, /, comparisons, sorting and numpy's
// pairwise summation, all of which are reproducible operation by operation (no FMA contraction: see the pragma below).
#include "hn_common.h"
#include "../../include/horizonnet_hip.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int MAXC = HN_FIT_MAX_CORNERS;

// numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum; np.sum / np.mean of a contiguous
// 1-D float64 array is exactly this -- checked against numpy on lengths 1..1024)
double pw_sum(const double* a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pw_sum(a, n2) + pw_sum(a + n2, n - n2);
}

inline double np_mean(const double* a, long n) { return pw_sum(a, n) / (double)n; }

struct VoteRes {
    double fit, p, l1;
    bool bad;      // the reference's `assert best_j > best_i` would have fired
};

// post_proc.py:75-98 through horizonnet_amd/postproc.py vote(): vec is consumed (sorted in place), tmp has room for n doubles
VoteRes vote(double* vec, long n, double tol, double* tmp)
{
    std::sort(vec, vec + n);
    const long L = n;
    long best_span = -1, best_i = -1, best_j = -1;
    long j = 0;
    for (long i = 0; i < L; ++i) {                                     // hn_vote_scan's loop (engine.hip)
        if (j < i) j = i;
        while (j + 1 < L && !((vec[j + 1] - vec[i]) + 1e-9 > tol)) ++j;
        if (j == i && 0.0 > tol) continue;
        const long span = j - i + 1;
        if (!((double)span < (double)L * 0.4) && span > best_span) {
            best_span = span;
            best_i = i;
            best_j = j;
        }
    }
    VoteRes r;
    r.bad = false;
    if (best_span < 0 || (double)L < tol) {
        r.fit = (L % 2) ? vec[L / 2] : (vec[L / 2 - 1] + vec[L / 2]) / 2.0;      // np.median of the sorted samples
        r.p = 0.0;
    } else {
        if (!(best_j > best_i)) r.bad = true;
        r.fit = np_mean(vec + best_i, best_j - best_i + 1);
        r.p = (double)(best_j - best_i + 1) / (double)L;
    }
    for (long i = 0; i < L; ++i) tmp[i] = vec[i] - r.fit < 0 ? -(vec[i] - r.fit) : vec[i] - r.fit;   // np.abs(vec - best_fit)
    r.l1 = np_mean(tmp, L);
    return r;
}

struct Wall {
    int type;
    double val, score;
    int gpid;
    int u0, u1;        // COLUMN index of the corner rays (table lookups of sin / cos); -1 = the reference's default u = -1 (entry W)
    bool tbd;
};

struct Ctx {
    const double* xs;          // [W] floor-plan x of the ceiling boundary per column
    const double* ys;
    const double* sin_u;       // [W + 1]
    const double* cos_u;
    int W;
    std::vector<int> gpid;     // [W]
    std::vector<double> bufx, bufy, tmp;
    bool bad = false;
    bool overflow = false;     // more walls than the fixed-size output of one panorama holds (HN_FIT_MAX_CORNERS): NOT an error of the layout
};

// _vote_axis: is the segment a constant-x (0) or constant-y (1) wall?
void vote_axis(Ctx& c, int g, double tol, int* axis, double* val, double* score)
{
    long n = 0;
    for (int k = 0; k < c.W; ++k)
        if (c.gpid[k] == g) {
            c.bufx[n] = c.xs[k];
            c.bufy[n] = c.ys[k];
            ++n;
        }
    if (n == 0) { c.bad = true; *axis = 0; *val = 0; *score = 0; return; }
    const VoteRes vx = vote(c.bufx.data(), n, tol, c.tmp.data());
    const VoteRes vy = vote(c.bufy.data(), n, tol, c.tmp.data());
    if (vx.bad || vy.bad) c.bad = true;
    // (sx, -lx) > (sy, -ly) as Python tuples
    const bool x_wins = vx.p != vy.p ? vx.p > vy.p : -vx.l1 > -vy.l1;
    if (x_wins) { *axis = 0; *val = vx.fit; *score = vx.p; }
    else { *axis = 1; *val = vy.fit; *score = vy.p; }
}

inline double sin_of(const Ctx& c, int col) { return c.sin_u[col < 0 ? c.W : col]; }
inline double cos_of(const Ctx& c, int col) { return c.cos_u[col < 0 ? c.W : col]; }

// corner_wall(src, u_key): the wall perpendicular to src through the point where src meets the corner ray
void corner_wall(const Ctx& c, const Wall& src, int ucol, int* axis, double* val)
{
    const double fw = 1024.0 / 2, fh = 512.0 / 2;          // floorW / 2, floorH / 2 (the reference's defaults; W x W/2 plan)
    if (src.type == 0) {                                   // y_where_ray_meets_x
        const double cc = (src.val - fw + 0.5) / sin_of(c, ucol);
        *axis = 1;
        *val = -cc * cos_of(c, ucol) + fh - 0.5;
    } else {                                               // x_where_ray_meets_y
        const double cc = -(src.val - fh + 0.5) / cos_of(c, ucol);
        *axis = 0;
        *val = cc * sin_of(c, ucol) + fw - 0.5;
    }
}

// wall_groups (get_gpid): column -> wall segment; the segment that wraps around the border is 0.  Returns the number of groups.
int wall_groups(Ctx& c, const int* peaks, int npk)
{
    std::fill(c.gpid.begin(), c.gpid.end(), 0);
    for (int i = 0; i < npk; ++i) c.gpid[peaks[i]] = 1;
    int run = 0;
    for (int k = 0; k < c.W; ++k) {
        run += c.gpid[k];
        c.gpid[k] = run;
    }
    const int last = c.gpid[c.W - 1];
    for (int k = 0; k < c.W; ++k)
        if (c.gpid[k] == last) c.gpid[k] = 0;
    // number of distinct ids = max id + 1 when every id below `last` occurs (it does: ids are a running count)
    int mx = 0;
    for (int k = 0; k < c.W; ++k) mx = std::max(mx, c.gpid[k]);
    return mx + 1;
}

bool walls_cuboid(Ctx& c, double tol, std::vector<Wall>& walls, int ngroups)
{
    if (ngroups != 4) return false;                        // assert len(np.unique(gpid)) == 4
    walls.clear();
    for (int j = 0; j < 4; ++j) {
        Wall w{};
        vote_axis(c, j, tol, &w.type, &w.val, &w.score);
        walls.push_back(w);
    }
    double balance[2] = {0, 0};
    for (int j = 0; j < 4; ++j) balance[j % 2] += walls[j].type == 0 ? walls[j].score : -walls[j].score;
    const int first = balance[0] > balance[1] ? 0 : 1;
    for (int j = 0; j < 4; ++j) walls[j].type = (first + j) % 2;
    return true;
}

bool walls_general(Ctx& c, const int* peaks, int n, double tol, std::vector<Wall>& walls, int ngroups)
{
    if (n != ngroups) return false;                        // assert n == len(np.unique(gpid))
    walls.clear();
    for (int j = 0; j < n; ++j) {
        Wall w{};
        vote_axis(c, j, tol, &w.type, &w.val, &w.score);
        w.gpid = j;
        w.u0 = peaks[(j - 1 + n) % n];
        w.u1 = peaks[j];
        w.tbd = true;
        walls.push_back(w);
    }
    for (;;) {
        int cur = -1;
        for (int i = 0; i < (int)walls.size(); ++i)
            if (walls[i].tbd && (cur == -1 || walls[i].score > walls[cur].score)) cur = i;
        if (cur == -1) break;
        if ((int)walls.size() > MAXC) { c.overflow = true; return false; }
        Wall& w = walls[cur];
        w.tbd = false;
        const int nw = (int)walls.size();
        const int pi = (cur - 1 + nw) % nw, ni = (cur + 1) % nw;
        const Wall prv = walls[pi], nxt = walls[ni];       // copies: the vector may be edited below
        const int open = (prv.tbd ? 1 : 0) + (nxt.tbd ? 1 : 0);
        if (open == 2) continue;
        if (open == 1) {
            const bool clash = (!prv.tbd && prv.type == w.type) || (!nxt.tbd && nxt.type == w.type);
            if (clash) {
                if (w.score >= -1) {
                    w.tbd = true;                          // decide it later, after its other neighbour
                    w.score -= 100;
                } else {                                   // second visit: force a perpendicular wall next to the settled neighbour
                    Wall f{};
                    f.score = 0; f.gpid = -1; f.u0 = -1; f.u1 = -1; f.tbd = false;
                    int at;
                    if (!prv.tbd) { at = cur; corner_wall(c, prv, prv.u1, &f.type, &f.val); }
                    else { at = ni; corner_wall(c, nxt, nxt.u0, &f.type, &f.val); }
                    walls.insert(walls.begin() + at, f);
                }
            }
            continue;
        }
        // both neighbours settled
        if (prv.type == nxt.type) {
            if (w.type == prv.type) {                      // three parallel walls in a row: turn the middle one
                w.type = (w.type + 1) % 2;
                long m = 0;
                const double* src = w.type == 0 ? c.xs : c.ys;
                for (int k = 0; k < c.W; ++k)
                    if (c.gpid[k] == w.gpid) c.bufx[m++] = src[k];
                if (m == 0) return false;
                w.val = np_mean(c.bufx.data(), m);
            }
        } else {                                           // neighbours perpendicular to each other: two walls meeting at a corner
            Wall a{}, b{};
            int dummy;
            a.type = nxt.type; corner_wall(c, prv, prv.u1, &dummy, &a.val);
            b.type = prv.type; corner_wall(c, nxt, nxt.u0, &dummy, &b.val);
            a.score = b.score = 0; a.gpid = b.gpid = -1; a.u0 = a.u1 = b.u0 = b.u1 = -1; a.tbd = b.tbd = false;
            walls[cur] = a;
            walls.insert(walls.begin() + cur + 1, b);
        }
    }
    if ((int)walls.size() > MAXC) { c.overflow = true; return false; }
    return true;
}

// polygon_is_simple (the stand-in for shapely's Polygon.is_valid at inference.py:120) on the float32-rounded plan
bool polygon_is_simple(const double* px_in, const double* py_in, int n_in)
{
    // repeated consecutive points (cyclically) are dropped first, as in postproc.polygon_is_simple
    std::vector<double> ux, uy;
    for (int i = 0; i < n_in; ++i) {
        const int pv = (i - 1 + n_in) % n_in;
        if (n_in > 1 && px_in[i] == px_in[pv] && py_in[i] == py_in[pv]) continue;
        ux.push_back(px_in[i]);
        uy.push_back(py_in[i]);
    }
    const double* px = ux.data();
    const double* py = uy.data();
    const int n = (int)ux.size();
    if (n < 3) return false;
    std::vector<double> qx(n), qy(n), ex(n), ey(n);
    for (int i = 0; i < n; ++i) {
        qx[i] = px[(i + 1) % n];
        qy[i] = py[(i + 1) % n];
    }
    double d1 = 0, d2 = 0;                                  // np.dot(p[:,0], q[:,1]) - np.dot(p[:,1], q[:,0])
    for (int i = 0; i < n; ++i) d1 += px[i] * qy[i];
    for (int i = 0; i < n; ++i) d2 += py[i] * qx[i];
    if (d1 - d2 == 0) return false;
    for (int i = 0; i < n; ++i) {
        ex[i] = qx[i] - px[i];
        ey[i] = qy[i] - py[i];
    }
    auto sgn = [](double v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
    auto turn = [&](int i, double cx, double cy) {          // sign of cross(edge_i, c - p_i)
        const double wx = cx - px[i], wy = cy - py[i];
        return sgn(ex[i] * wy - ey[i] * wx);
    };
    auto boxed = [&](int i, double cx, double cy) {
        const double lox = std::min(px[i], qx[i]), hix = std::max(px[i], qx[i]);
        const double loy = std::min(py[i], qy[i]), hiy = std::max(py[i], qy[i]);
        return lox <= cx && cx <= hix && loy <= cy && cy <= hiy;
    };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int gap = ((j - i) % n + n) % n;
            if (!(gap >= 2 && gap <= n - 2)) continue;
            const int sp_ij = turn(i, px[j], py[j]), sq_ij = turn(i, qx[j], qy[j]);
            const int sp_ji = turn(j, px[i], py[i]), sq_ji = turn(j, qx[i], qy[i]);
            const bool cross = (sp_ij != sq_ij) && (sp_ji != sq_ji);
            const bool touch_ij = (sp_ij == 0 && boxed(i, px[j], py[j])) || (sq_ij == 0 && boxed(i, qx[j], qy[j]));
            const bool touch_ji = (sp_ji == 0 && boxed(j, px[i], py[i])) || (sq_ji == 0 && boxed(j, qx[i], qy[i]));
            if (cross || touch_ij || touch_ji) return false;
        }
    return true;
}

// peaks of a mask in column order
int collect_peaks(const unsigned char* mask, int W, std::vector<int>& pk)
{
    pk.clear();
    for (int k = 0; k < W; ++k)
        if (mask[k]) pk.push_back(k);
    return (int)pk.size();
}

int fit_one(Ctx& c, const unsigned char* mask, double tol, int force_cuboid, double* pts, int32_t* npts)
{
    std::vector<int> pk;
    std::vector<Wall> walls;
    collect_peaks(mask, c.W, pk);
    const int ng = wall_groups(c, pk.data(), (int)pk.size());
    bool ok;
    if (force_cuboid) {
        ok = walls_cuboid(c, tol, walls, ng);
    } else {
        ok = !pk.empty() && walls_general(c, pk.data(), (int)pk.size(), tol, walls, ng);
        if (ok && !c.bad) {
            // inference.py:113-120: the float32 plan of the wall values, checked for validity
            const int n = (int)walls.size();
            std::vector<double> px(n, 0.0), py(n, 0.0);
            for (int i = 0; i < n; ++i) {
                const Wall& w = walls[i];
                const Wall& pv = walls[(i - 1 + n) % n];
                double* cur[2] = {&px[i], &py[i]};
                *cur[w.type] = (double)(float)w.val;
                *cur[pv.type] = (double)(float)pv.val;
            }
            if (!polygon_is_simple(px.data(), py.data(), n)) {
                *npts = 0;
                return 1;                                  // "Fail to generate valid general layout!!": the caller falls back to the cuboid fit
            }
        }
    }
    if (c.overflow || (int)walls.size() > MAXC) {
        *npts = 0;
        return 3;                                          // a valid layout with more walls than this call's output rows: the caller re-runs it
    }
    if (!ok || c.bad || walls.empty()) {
        *npts = 0;
        return 2;                                          // one of the reference's assertions would have fired
    }
    const int n = (int)walls.size();
    for (int j = 0; j < n; ++j) {
        const Wall& w = walls[j];
        const Wall& nx = walls[(j + 1) % n];
        if (w.type == 1) { pts[2 * j] = nx.val; pts[2 * j + 1] = w.val; }
        else { pts[2 * j] = w.val; pts[2 * j + 1] = nx.val; }
    }
    *npts = n;
    return 0;
}

// numpy's pairwise summation in float32 (FLOAT_pairwise_sum): what np.mean of a contiguous float32 vector sums with
float pw_sum_f32(const float* a, long n)
{
    if (n < 8) {
        float res = 0.f;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pw_sum_f32(a, n2) + pw_sum_f32(a + n2, n - n2);
}

}  // namespace

// mean_percentile (post_proc.py:69-72) for B rows: out[b] = mean of the z[b][k] with lo[b] <= z <= hi[b], float32 like
// numpy's vec[(lo <= vec) & (vec <= hi)].mean() on a float32 vector (compaction in order, pairwise sum, one division)
extern "C" int hn_interquartile_mean_f32(const float* z, const float* lo, const float* hi, int B, int W, float* out)
{
    HN_REQUIRE(z && lo && hi && out && B >= 0 && W >= 1, "hn_interquartile_mean_f32: bad argument");
    std::vector<float> buf(W);
    for (int b = 0; b < B; ++b) {
        const float* row = z + (size_t)b * W;
        long n = 0;
        for (int k = 0; k < W; ++k)
            if (lo[b] <= row[k] && row[k] <= hi[b]) buf[n++] = row[k];
        out[b] = n ? pw_sum_f32(buf.data(), n) / (float)n : (0.f / 0.f);       // numpy: mean of an empty slice = nan
    }
    return 0;
}

extern "C" int hn_layout_fit_batch(const double* xs, const double* ys, const unsigned char* peak_mask, const double* sin_u,
                                   const double* cos_u, const double* tol, int B, int W, int force_cuboid, int threads, double* pts,
                                   int32_t* npts, int32_t* flags)
{
    HN_REQUIRE(xs && ys && peak_mask && sin_u && cos_u && tol && pts && npts && flags, "hn_layout_fit_batch: null pointer");
    HN_REQUIRE(B >= 0 && W >= 8 && W <= 65536, "hn_layout_fit_batch: bad B=%d / W=%d", B, W);
    HN_REQUIRE(W == 1024, "hn_layout_fit_batch: the floor-plan constants follow the reference's 1024 x 512 panorama (W=%d)", W);
    if (B == 0) return 0;
    int nt = threads < 1 ? 1 : threads;
    if (nt > B) nt = B;
    std::atomic<int> next(0);
    auto work = [&]() {
        Ctx c;
        c.W = W;
        c.sin_u = sin_u;
        c.cos_u = cos_u;
        c.gpid.resize(W);
        c.bufx.resize(W);
        c.bufy.resize(W);
        c.tmp.resize(W);
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            c.xs = xs + (size_t)b * W;
            c.ys = ys + (size_t)b * W;
            c.bad = false;
            c.overflow = false;
            flags[b] = fit_one(c, peak_mask + (size_t)b * W, tol[b], force_cuboid, pts + (size_t)b * MAXC * 2, npts + b);
        }
    };
    if (nt == 1) {
        work();
        return 0;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int i = 1; i < nt; ++i) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return 0;
}
