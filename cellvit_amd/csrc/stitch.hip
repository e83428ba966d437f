// Slide-level de-duplication of margin cells on the device (reference CellPostProcessor._remove_overlap,
// cell_segmentation/inference/cell_detection.py:676-767).  The reference builds a shapely STRtree over the cells' contour
// polygons and, per cell, intersects it with every polygon whose envelope meets its own; here
//   1. candidate pairs = cells whose GLOBAL bounding boxes overlap, found with a uniform 64-px grid (count / scan / fill /
//      emit: every pair is emitted exactly once, by the grid cell that contains the top-left corner of the boxes' intersection);
//   2. per cell the exact polygon area (integer shoelace), per candidate pair the EXACT area of the intersection of the two
//      contour polygons (even-odd interiors) by slab decomposition — one thread per pair;
//   3. the greedy rounds of the reference ("of every group of cells overlapping by > 1 % of either area the largest OTHER cell
//      survives", up to 20 rounds) run on the host over the pair list (cv_stitch_select): they are a sequential sweep in cell
//      order by definition, a few ms for 3e5 cells.
// Contours are the (x, y) integer points of the per-tile contour tracer in slide coordinates.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/cellvit_amd.h"

void cva_set_error(const char* fmt, ...);

namespace {

constexpr int GRID_PX = 64;           // grid cell size in pixels (a nucleus bounding box covers 1-4 cells)
constexpr int MAX_ACT = 24;           // active edges of one polygon inside a unit slab (a simple nucleus outline has 2-6)
constexpr int MAX_EV = 96;            // edge crossings inside a unit slab

struct Grid { int gy0, gx0, ny, nx; };

__device__ __forceinline__ int gdiv(int v) { return v >= 0 ? v / GRID_PX : -((-v + GRID_PX - 1) / GRID_PX); }   // floor division

// bbox: [n, 4] = (r0, c0, r1, c1), rows / columns in slide coordinates
__global__ void k_grid_count(const int32_t* __restrict__ bbox, int n, Grid g, int32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y0 = gdiv(bbox[4 * i]) - g.gy0, x0 = gdiv(bbox[4 * i + 1]) - g.gx0;
    const int y1 = gdiv(bbox[4 * i + 2]) - g.gy0, x1 = gdiv(bbox[4 * i + 3]) - g.gx0;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) atomicAdd(&count[(long)y * g.nx + x], 1);
}

__global__ void k_grid_fill(const int32_t* __restrict__ bbox, int n, Grid g, const int32_t* __restrict__ start,
                            int32_t* __restrict__ cursor, int32_t* __restrict__ entry) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y0 = gdiv(bbox[4 * i]) - g.gy0, x0 = gdiv(bbox[4 * i + 1]) - g.gx0;
    const int y1 = gdiv(bbox[4 * i + 2]) - g.gy0, x1 = gdiv(bbox[4 * i + 3]) - g.gx0;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const long c = (long)y * g.nx + x;
            entry[start[c] + atomicAdd(&cursor[c], 1)] = i;
        }
}

// Pairs (i < j) with strictly overlapping boxes (touching boxes have no common area: stitch rule `a0 >= r1 ... -> skip`).
__global__ void k_grid_pairs(const int32_t* __restrict__ bbox, int n, Grid g, const int32_t* __restrict__ start,
                             const int32_t* __restrict__ entry, int32_t* __restrict__ pairs, int cap, int32_t* __restrict__ n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r0 = bbox[4 * i], c0 = bbox[4 * i + 1], r1 = bbox[4 * i + 2], c1 = bbox[4 * i + 3];
    const int y0 = gdiv(r0) - g.gy0, x0 = gdiv(c0) - g.gx0, y1 = gdiv(r1) - g.gy0, x1 = gdiv(c1) - g.gx0;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const long c = (long)y * g.nx + x;
            for (int e = start[c]; e < start[c + 1]; ++e) {
                const int j = entry[e];
                if (j <= i) continue;
                const int a0 = bbox[4 * j], b0 = bbox[4 * j + 1], a1 = bbox[4 * j + 2], b1 = bbox[4 * j + 3];
                if (a0 >= r1 || a1 <= r0 || b0 >= c1 || b1 <= c0) continue;
                // the pair belongs to the grid cell that holds the top-left corner of the boxes' intersection
                if (gdiv(max(r0, a0)) - g.gy0 != y || gdiv(max(c0, b0)) - g.gx0 != x) continue;
                const int k = atomicAdd(n_pairs, 1);
                if (k < cap) { pairs[2 * k] = i; pairs[2 * k + 1] = j; }
            }
        }
}

// Rings that are not simple (reference: `if not poly.is_valid` -> Polygon.buffer(0) -> largest part, cell_detection.py:689-704).
// The contours are outer borders traced on the pixel lattice: they never cross themselves but can TOUCH themselves (a blob that is
// 8-connected through a diagonal pinch passes the pinch pixel twice; a one-pixel-wide spur is walked out and back).  flags[i] = 1
// when two non-adjacent edges of ring i share a point, two consecutive edges fold back onto each other, or a vertex repeats —
// one thread per ring, O(edges^2) integer orientation tests (a nucleus outline has 10-60 edges).  The flagged rings (a handful
// per slide) are repaired by host code of the library (cv_stitch_repair_rings).
__device__ __forceinline__ long long orient2(int ax, int ay, int bx, int by, int cx, int cy) {
    return (long long)(bx - ax) * (cy - ay) - (long long)(by - ay) * (cx - ax);
}
__device__ __forceinline__ bool in_box(int ax, int ay, int bx, int by, int px, int py) {
    return px >= min(ax, bx) && px <= max(ax, bx) && py >= min(ay, by) && py <= max(ay, by);
}
__global__ void k_ring_flags(const int64_t* __restrict__ off, const int32_t* __restrict__ xy, int n, uint8_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long o = off[i];
    const int m = (int)(off[i + 1] - o);
    const int32_t* p = xy + 2 * o;
    bool bad = false;
    for (int a = 0; a < m && !bad; ++a) {
        const int a1 = a + 1 == m ? 0 : a + 1;
        const int px = p[2 * a], py = p[2 * a + 1], qx = p[2 * a1], qy = p[2 * a1 + 1];
        if (px == qx && py == qy) { bad = m > 1; break; }                 // repeated consecutive vertex
        {   // the next edge folds back onto this one (spur tip)
            const int a2 = a1 + 1 == m ? 0 : a1 + 1;
            const int rx = p[2 * a2], ry = p[2 * a2 + 1];
            if (m >= 3 && orient2(px, py, qx, qy, rx, ry) == 0 &&
                (long long)(qx - px) * (rx - qx) + (long long)(qy - py) * (ry - qy) < 0) { bad = true; break; }
        }
        for (int b = a + 2; b < m; ++b) {
            if (a == 0 && b == m - 1) continue;                          // adjacent through the closing edge
            const int b1 = b + 1 == m ? 0 : b + 1;
            const int rx = p[2 * b], ry = p[2 * b + 1], sx = p[2 * b1], sy = p[2 * b1 + 1];
            if (max(px, qx) < min(rx, sx) || max(rx, sx) < min(px, qx) || max(py, qy) < min(ry, sy) || max(ry, sy) < min(py, qy)) continue;
            const long long o1 = orient2(px, py, qx, qy, rx, ry), o2 = orient2(px, py, qx, qy, sx, sy);
            const long long o3 = orient2(rx, ry, sx, sy, px, py), o4 = orient2(rx, ry, sx, sy, qx, qy);
            if (((o1 > 0) != (o2 > 0)) && o1 != 0 && o2 != 0 && ((o3 > 0) != (o4 > 0)) && o3 != 0 && o4 != 0) { bad = true; break; }
            if ((o1 == 0 && in_box(px, py, qx, qy, rx, ry)) || (o2 == 0 && in_box(px, py, qx, qy, sx, sy)) ||
                (o3 == 0 && in_box(rx, ry, sx, sy, px, py)) || (o4 == 0 && in_box(rx, ry, sx, sy, qx, qy))) { bad = true; break; }
        }
    }
    flags[i] = bad ? 1 : 0;
}

// 2 * signed area of the closed polygon through the contour points: integer arithmetic, exact
__global__ void k_poly_area(const int64_t* __restrict__ off, const int32_t* __restrict__ xy, int n, double* __restrict__ area) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long o = off[i];
    const int m = (int)(off[i + 1] - o);
    long long s = 0;
    if (m >= 3) {
        const int32_t* p = xy + 2 * o;
        long long xp = p[2 * (m - 1)], yp = p[2 * (m - 1) + 1];
        for (int k = 0; k < m; ++k) {
            const long long x = p[2 * k], y = p[2 * k + 1];
            s += xp * y - yp * x;
            xp = x; yp = y;
        }
    }
    area[i] = 0.5 * (double)(s < 0 ? -s : s);
}

// active edges of polygon p inside the unit slab [y, y+1]: x at ordinate y and dx/dy.  Vertex ordinates are integers, so an edge
// either spans the whole slab or misses its interior; horizontal edges never qualify.  Coordinates are taken relative to (ox, oy).
__device__ __forceinline__ int active_edges(const int32_t* p, int m, int y, int ox, double* xs, double* sl) {
    int cnt = 0;
    int xp = p[2 * (m - 1)] - ox, yp = p[2 * (m - 1) + 1];
    for (int k = 0; k < m; ++k) {
        const int x = p[2 * k] - ox, yy = p[2 * k + 1];
        const int lo = min(yp, yy), hi = max(yp, yy);
        if (lo <= y && hi >= y + 1) {
            if (cnt < MAX_ACT) {
                const double s = (double)(x - xp) / (double)(yy - yp);
                xs[cnt] = (double)xp + s * (double)(y - yp);
                sl[cnt] = s;
            }
            ++cnt;
        }
        xp = x; yp = yy;
    }
    return cnt;
}

__device__ __forceinline__ void sort_small(double* v, int n) {
    for (int a = 1; a < n; ++a) {
        const double t = v[a];
        int b = a - 1;
        while (b >= 0 && v[b] > t) { v[b + 1] = v[b]; --b; }
        v[b + 1] = t;
    }
}

// EXACT area of the intersection of two polygons (even-odd interiors) by slab decomposition: between two consecutive event
// ordinates (the integers — every vertex ordinate is one — and the crossings of an edge of A with an edge of B) each interval
// end point is linear in y, so the common length L(y) is linear and the midpoint rule integrates it exactly.
// inter[k] = area, or -1 when a slab exceeded the fixed per-thread capacities (the caller evaluates those pairs on the host).
__global__ void k_pair_inter(const int32_t* __restrict__ pairs, int n_pairs, const int64_t* __restrict__ off,
                             const int32_t* __restrict__ xy, double* __restrict__ inter) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const int i = pairs[2 * k], j = pairs[2 * k + 1];
    const int32_t* pa = xy + 2 * off[i];
    const int32_t* pb = xy + 2 * off[j];
    const int na = (int)(off[i + 1] - off[i]), nb = (int)(off[j + 1] - off[j]);
    if (na < 3 || nb < 3) { inter[k] = 0.0; return; }
    int ylo_a = pa[1], yhi_a = pa[1], ylo_b = pb[1], yhi_b = pb[1], xmin = pa[0];
    for (int q = 1; q < na; ++q) { ylo_a = min(ylo_a, pa[2 * q + 1]); yhi_a = max(yhi_a, pa[2 * q + 1]); xmin = min(xmin, pa[2 * q]); }
    for (int q = 0; q < nb; ++q) { ylo_b = min(ylo_b, pb[2 * q + 1]); yhi_b = max(yhi_b, pb[2 * q + 1]); xmin = min(xmin, pb[2 * q]); }
    const int ylo = max(ylo_a, ylo_b), yhi = min(yhi_a, yhi_b);
    double total = 0.0;
    double xa[MAX_ACT], sa[MAX_ACT], xb[MAX_ACT], sb[MAX_ACT], ev[MAX_EV + 2], ta[MAX_ACT], tb[MAX_ACT];
    for (int y = ylo; y < yhi; ++y) {
        const int ca = active_edges(pa, na, y, xmin, xa, sa);
        if (ca == 0) continue;
        const int cb = active_edges(pb, nb, y, xmin, xb, sb);
        if (cb == 0) continue;
        if (ca > MAX_ACT || cb > MAX_ACT) { inter[k] = -1.0; return; }
        int ne = 0;
        ev[ne++] = 0.0;
        for (int a = 0; a < ca; ++a)
            for (int b = 0; b < cb; ++b) {
                const double ds = sa[a] - sb[b];
                if (ds == 0.0) continue;
                const double t = (xb[b] - xa[a]) / ds;            // ordinate of the crossing, relative to y
                if (t > 0.0 && t < 1.0) {
                    if (ne >= MAX_EV + 1) { inter[k] = -1.0; return; }
                    ev[ne++] = t;
                }
            }
        ev[ne++] = 1.0;
        sort_small(ev, ne);
        for (int e = 0; e + 1 < ne; ++e) {
            const double t0 = ev[e], t1 = ev[e + 1];
            if (t1 <= t0) continue;
            const double tm = 0.5 * (t0 + t1);
            for (int a = 0; a < ca; ++a) ta[a] = xa[a] + sa[a] * tm;
            for (int b = 0; b < cb; ++b) tb[b] = xb[b] + sb[b] * tm;
            sort_small(ta, ca);
            sort_small(tb, cb);
            double len = 0.0;
            for (int a = 0; a + 1 < ca; a += 2)
                for (int b = 0; b + 1 < cb; b += 2) {
                    const double l = fmin(ta[a + 1], tb[b + 1]) - fmax(ta[a], tb[b]);
                    if (l > 0.0) len += l;
                }
            total += len * (t1 - t0);
        }
    }
    inter[k] = total;
}

}  // namespace

#define ST_CHECK(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess) { cva_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); rc = CV_ERR_HIP; goto done; } \
    } while (0)

extern "C" int cv_stitch_overlaps(const int32_t* bbox, const int64_t* ct_off, const int32_t* ct_xy, int n, const int32_t* extent,
                                  int32_t* pairs, double* inter, double* area, int cap, int32_t* n_pairs_host, void* stream_) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    int rc = CV_OK;
    int32_t *count = nullptr, *start = nullptr, *cursor = nullptr, *entry = nullptr, *np_dev = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    int32_t total_entries = 0, np = 0;
    *n_pairs_host = 0;
    if (n <= 0) return CV_OK;
    if (!bbox || !ct_off || !ct_xy || !extent || !pairs || !inter || !area || cap <= 0) { cva_set_error("cv_stitch_overlaps: bad argument"); return CV_ERR_INVALID; }
    {
        Grid g;
        auto fdiv = [](int v) { return v >= 0 ? v / GRID_PX : -((-v + GRID_PX - 1) / GRID_PX); };
        g.gy0 = fdiv(extent[0]); g.gx0 = fdiv(extent[1]);
        g.ny = fdiv(extent[2]) - g.gy0 + 1; g.nx = fdiv(extent[3]) - g.gx0 + 1;
        const long cells = (long)g.ny * g.nx;
        if (g.ny <= 0 || g.nx <= 0 || cells > (1L << 30)) { cva_set_error("cv_stitch_overlaps: bad extent"); return CV_ERR_INVALID; }
        const int T = 256, nb = (n + T - 1) / T;
        ST_CHECK(hipMallocAsync((void**)&count, (cells + 1) * 4, st));
        ST_CHECK(hipMallocAsync((void**)&start, (cells + 1) * 4, st));
        ST_CHECK(hipMallocAsync((void**)&cursor, (cells + 1) * 4, st));
        ST_CHECK(hipMallocAsync((void**)&np_dev, 4, st));
        ST_CHECK(hipMemsetAsync(count, 0, (cells + 1) * 4, st));
        ST_CHECK(hipMemsetAsync(cursor, 0, (cells + 1) * 4, st));
        ST_CHECK(hipMemsetAsync(np_dev, 0, 4, st));
        hipLaunchKernelGGL(k_grid_count, dim3(nb), dim3(T), 0, st, bbox, n, g, count);
        ST_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, count, start, (int)(cells + 1), st));
        ST_CHECK(hipMallocAsync(&tmp, tmp_bytes ? tmp_bytes : 16, st));
        ST_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, count, start, (int)(cells + 1), st));
        ST_CHECK(hipMemcpyAsync(&total_entries, start + cells, 4, hipMemcpyDeviceToHost, st));
        ST_CHECK(hipStreamSynchronize(st));
        ST_CHECK(hipMallocAsync((void**)&entry, (size_t)(total_entries > 0 ? total_entries : 1) * 4, st));
        hipLaunchKernelGGL(k_grid_fill, dim3(nb), dim3(T), 0, st, bbox, n, g, start, cursor, entry);
        hipLaunchKernelGGL(k_grid_pairs, dim3(nb), dim3(T), 0, st, bbox, n, g, start, entry, pairs, cap, np_dev);
        hipLaunchKernelGGL(k_poly_area, dim3(nb), dim3(T), 0, st, ct_off, ct_xy, n, area);
        ST_CHECK(hipMemcpyAsync(&np, np_dev, 4, hipMemcpyDeviceToHost, st));
        ST_CHECK(hipStreamSynchronize(st));
        if (np > cap) {      // the caller retries with the count returned here
            *n_pairs_host = np;
            cva_set_error("cv_stitch_overlaps: %d candidate pairs exceed the capacity %d", np, cap); rc = CV_ERR_SHAPE; goto done;
        }
        if (np > 0) hipLaunchKernelGGL(k_pair_inter, dim3((np + 63) / 64), dim3(64), 0, st, pairs, np, ct_off, ct_xy, inter);
        ST_CHECK(hipGetLastError());
        ST_CHECK(hipStreamSynchronize(st));
        *n_pairs_host = np;
    }
done:
    if (count) (void)hipFreeAsync(count, st);
    if (start) (void)hipFreeAsync(start, st);
    if (cursor) (void)hipFreeAsync(cursor, st);
    if (entry) (void)hipFreeAsync(entry, st);
    if (np_dev) (void)hipFreeAsync(np_dev, st);
    if (tmp) (void)hipFreeAsync(tmp, st);
    return rc;
}

extern "C" int cv_stitch_ring_flags(const int64_t* ct_off, const int32_t* ct_xy, int n, uint8_t* flags, void* stream_) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    if (n <= 0) return CV_OK;
    if (!ct_off || !ct_xy || !flags) { cva_set_error("cv_stitch_ring_flags: bad argument"); return CV_ERR_INVALID; }
    hipLaunchKernelGGL(k_ring_flags, dim3((n + 63) / 64), dim3(64), 0, st, ct_off, ct_xy, n, flags);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { cva_set_error("cv_stitch_ring_flags: %s", hipGetErrorString(e)); return CV_ERR_HIP; }
    return CV_OK;
}

// `Polygon.buffer(0)` -> largest part (cell_detection.py:689-704) on lattice rings — HOST code.  A ring whose lattice chain (every
// lattice point along its edges; edges of a traced border run along the 8 lattice directions) visits a point twice is cut into
// loops at the repeated points — the path between the two visits is a closed loop, what remains closes through the ring's first
// point — and replaced by the loop of the largest |area| (the first such loop closed along the ring when areas tie; zero-width
// spurs are loops of area 0 and vanish).  flags (u8 [n], from cv_stitch_ring_flags) limits the work to the flagged rings; NULL =
// examine every ring.  out_off i64 [n+1] / out_xy i32 [>= ct_off[n], 2]: all rings, repaired ones with collinear points removed
// (a lobe never has more vertices than its ring).  Edges that do not run along a lattice direction are kept as they are.
namespace {
struct Pt { int32_t x, y; };
inline bool pt_eq(const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }

bool largest_lobe(const int32_t* p, int m, std::vector<Pt>& chain, std::vector<Pt>& best) {
    chain.clear();
    for (int k = 0; k < m; ++k) {
        const int k1 = k + 1 == m ? 0 : k + 1;
        const int x0 = p[2 * k], y0 = p[2 * k + 1], dx = p[2 * k1] - x0, dy = p[2 * k1 + 1] - y0;
        const int steps = std::max(std::abs(dx), std::abs(dy));
        if (steps == 0) continue;
        if (!(dx == 0 || dy == 0 || std::abs(dx) == std::abs(dy))) { chain.push_back({x0, y0}); continue; }
        const int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0);
        for (int t = 0; t < steps; ++t) chain.push_back({x0 + t * sx, y0 + t * sy});
    }
    const int L = (int)chain.size();
    // first-visit position of every lattice point on the current stack: sorted index of (x, y) -> slot
    std::vector<int> order(L), slot_of(L), first(L, -1);
    for (int i = 0; i < L; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        return chain[a].x != chain[b].x ? chain[a].x < chain[b].x : (chain[a].y != chain[b].y ? chain[a].y < chain[b].y : a < b);
    });
    bool repeated = false;
    int ids = 0;
    for (int r = 0; r < L; ++r) {
        if (r > 0 && pt_eq(chain[order[r]], chain[order[r - 1]])) { slot_of[order[r]] = slot_of[order[r - 1]]; repeated = true; }
        else slot_of[order[r]] = ids++;
    }
    if (!repeated) return false;
    std::vector<int> stack;            // chain indices
    std::vector<Pt> loop;
    long long best_a = -1;
    best.clear();
    auto consider = [&](const std::vector<Pt>& lp) {
        long long s2 = 0;
        const int c = (int)lp.size();
        if (c >= 3)
            for (int k = 0; k < c; ++k) { const Pt& a = lp[k]; const Pt& b = lp[k + 1 == c ? 0 : k + 1]; s2 += (long long)a.x * b.y - (long long)b.x * a.y; }
        if (s2 < 0) s2 = -s2;
        if (s2 > best_a) { best_a = s2; best = lp; }
    };
    for (int i = 0; i < L; ++i) {
        const int id = slot_of[i];
        if (first[id] >= 0) {
            const int k = first[id];
            loop.clear();
            for (int q = k; q < (int)stack.size(); ++q) loop.push_back(chain[stack[q]]);
            for (int q = k + 1; q < (int)stack.size(); ++q) first[slot_of[stack[q]]] = -1;
            stack.resize(k + 1);
            consider(loop);
        } else {
            first[id] = (int)stack.size();
            stack.push_back(i);
        }
    }
    loop.clear();
    for (int q : stack) loop.push_back(chain[q]);
    consider(loop);
    // remove collinear points of the winning loop (keep direction changes)
    std::vector<Pt> simp;
    const int c = (int)best.size();
    for (int k = 0; k < c; ++k) {
        const Pt& a = best[(k + c - 1) % c]; const Pt& b = best[k]; const Pt& d = best[(k + 1) % c];
        const long long cr = (long long)(b.x - a.x) * (d.y - b.y) - (long long)(b.y - a.y) * (d.x - b.x);
        const long long dt = (long long)(b.x - a.x) * (d.x - b.x) + (long long)(b.y - a.y) * (d.y - b.y);
        if (cr != 0 || dt < 0) simp.push_back(b);
    }
    best.swap(simp);
    return true;
}
}  // namespace

extern "C" int cv_stitch_repair_rings(const int64_t* ct_off, const int32_t* ct_xy, int n, const uint8_t* flags, int64_t* out_off,
                                      int32_t* out_xy, int32_t* n_repaired) {
    if (n < 0 || (n && (!ct_off || !ct_xy || !out_off || !out_xy))) { cva_set_error("cv_stitch_repair_rings: bad argument"); return CV_ERR_INVALID; }
    std::vector<Pt> chain, best;
    int64_t w = 0;
    int rep = 0;
    if (out_off) out_off[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t o = ct_off[i];
        const int m = (int)(ct_off[i + 1] - o);
        const int32_t* p = ct_xy + 2 * o;
        if (m >= 3 && (!flags || flags[i]) && largest_lobe(p, m, chain, best)) {
            if ((int)best.size() > m) { cva_set_error("cv_stitch_repair_rings: lobe of ring %d has %d vertices, the ring %d", i, (int)best.size(), m); return CV_ERR_SHAPE; }
            for (const Pt& q : best) { out_xy[2 * w] = q.x; out_xy[2 * w + 1] = q.y; ++w; }
            ++rep;
        } else {
            for (int k = 0; k < 2 * m; ++k) out_xy[2 * w + k] = p[k];
            w += m;
        }
        out_off[i + 1] = w;
    }
    if (n_repaired) *n_repaired = rep;
    return CV_OK;
}

// The greedy rounds of CellPostProcessor._remove_overlap (cell_detection.py:676-767) over a pair list: pure host code.
//   pairs [n_pairs, 2] (i, j), overlap[k] != 0 when the pair overlaps by more than 1 % of either area, area [n], alive [n] in/out.
// Per round, cells are visited in index order; a visited cell i collects its not-yet-visited live overlap partners (in index
// order), marks them visited, and the round keeps the LARGEST of them (ties: the FIRST of equal areas in that order, as np.argmax
// over the submerger list, cell_detection.py:743-746) — or i itself when it has none.  Stops after a round without overlaps or
// after max_rounds.
extern "C" int cv_stitch_select(const int32_t* pairs, const uint8_t* overlap, int n_pairs, const double* area, uint8_t* alive,
                                int n, int max_rounds, int32_t* rounds_out, int32_t* overlaps_out) {
    if (n < 0 || n_pairs < 0 || (n_pairs && (!pairs || !overlap)) || (n && (!area || !alive))) { cva_set_error("cv_stitch_select: bad argument"); return CV_ERR_INVALID; }
    std::vector<int32_t> deg(n + 1, 0);
    for (int k = 0; k < n_pairs; ++k) {
        if (!overlap[k]) continue;
        const int i = pairs[2 * k], j = pairs[2 * k + 1];
        if (i < 0 || j < 0 || i >= n || j >= n || i == j) { cva_set_error("cv_stitch_select: pair %d out of range", k); return CV_ERR_INVALID; }
        ++deg[i + 1]; ++deg[j + 1];
    }
    for (int i = 0; i < n; ++i) deg[i + 1] += deg[i];
    std::vector<int32_t> adj(deg[n]), cur(deg.begin(), deg.end() - 1);
    for (int k = 0; k < n_pairs; ++k) {
        if (!overlap[k]) continue;
        const int i = pairs[2 * k], j = pairs[2 * k + 1];
        adj[cur[i]++] = j; adj[cur[j]++] = i;
    }
    for (int i = 0; i < n; ++i) std::sort(adj.begin() + deg[i], adj.begin() + deg[i + 1]);
    std::vector<uint8_t> done(n), next(n);
    int r = 0;
    for (; r < max_rounds; ++r) {
        std::fill(done.begin(), done.end(), 0);
        std::fill(next.begin(), next.end(), 0);
        int overlaps = 0;
        for (int i = 0; i < n; ++i) {
            if (!alive[i] || done[i]) continue;
            int best = -1;
            for (int e = deg[i]; e < deg[i + 1]; ++e) {
                const int j = adj[e];
                if (!alive[j] || done[j]) continue;
                ++overlaps;
                done[j] = 1;
                if (best < 0 || area[j] > area[best]) best = j;
            }
            next[best >= 0 ? best : i] = 1;
            done[i] = 1;
        }
        for (int i = 0; i < n; ++i) alive[i] = next[i];
        if (overlaps_out && r < max_rounds) overlaps_out[r] = overlaps;
        if (overlaps == 0) { ++r; break; }
    }
    if (rounds_out) *rounds_out = r;
    return CV_OK;
}
