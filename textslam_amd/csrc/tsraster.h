// tsraster.h -- cv::fillPoly scan conversion of one quad on the device, shared by the BA library (mu / sigma of a projected text
// box, text label image: tool::CalTextinfo, tool::TextBoxWithFill) and the frame front-end (tool::GetBoxAllPixs).
// OpenCV semantics restated: boundary with cv::LineIterator (8-connected, after cv::clipLine), interior with FillEdgeCollection
// (16.16 fixed-point scanline spans).
#pragma once
#include <hip/hip_runtime.h>
__device__ int clip_line_dev(long long Wd, long long Hd, long long &x1, long long &y1, long long &x2, long long &y2) {
    long long right = Wd - 1, bottom = Hd - 1;
    int c1 = (x1 < 0) + (x1 > right)*2 + (y1 < 0)*4 + (y1 > bottom)*8;
    int c2 = (x2 < 0) + (x2 > right)*2 + (y2 < 0)*4 + (y2 > bottom)*8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (long long)((double)(a - y1)*(double)(x2 - x1)/(double)(y2 - y1)); y1 = a; c1 = (x1 < 0) + (x1 > right)*2; }
        if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (long long)((double)(a - y2)*(double)(x2 - x1)/(double)(y2 - y1)); y2 = a; c2 = (x2 < 0) + (x2 > right)*2; }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) { a = c1 == 1 ? 0 : right; y1 += (long long)((double)(a - x1)*(double)(y2 - y1)/(double)(x2 - x1)); x1 = a; c1 = 0; }
            if (c2) { a = c2 == 1 ? 0 : right; y2 += (long long)((double)(a - x2)*(double)(y2 - y1)/(double)(x2 - x1)); x2 = a; c2 = 0; }
        }
    }
    return (c1 | c2) == 0;
}
// C's truncating num / den for |num| < 2^52, 0 < |den| < 2^32 (16.16 fixed-point slopes of edges between int pixel coordinates: |num| <= 2^48) by ONE fp64 division
// and an exact correction from the remainder: the compiler's expansion of a 64-bit integer division is a long-division loop of several hundred instructions,
// and a scanline fill does four of them per thread (round 6: the quad was 30 k of a mu / sigma workgroup's 55 k cycles).  The rounded quotient of two exactly
// represented integers is within 2^-5 of the true one here, so its truncation is off by at most one; the remainder says which way.
__device__ __forceinline__ long long div_trunc_small(long long num, long long den) {
    long long q = (long long)((double)num/(double)den);
    const long long r = num - q*den, ad = den < 0 ? -den : den;
    const long long rs = num < 0 ? -r : r;                     // the remainder must carry the sign of num: rs in [0, |den|)
    const long long sq = ((num < 0) != (den < 0)) ? -1 : 1;
    if (rs < 0) q -= sq; else if (rs >= ad) q += sq;
    return q;
}
#define MS_MASK_WORDS 9600        /* 640*480/32 bits */
// cv::fillPoly of one quad (integer corners s_xy, image w x hh) into an LDS bit mask: boundary lines with cv::LineIterator
// (8-connected) by 4 threads, interior by FillEdgeCollection scanlines (16.16 fixed point), one thread per row.  The caller
// clears the mask and synchronises before and after.
__device__ __forceinline__ void raster_quad(unsigned *mask, const int *s_xy, int w, int hh, int tid, int nthreads, long long *rq_dbg = nullptr) {
    const long long rq_t0 = rq_dbg ? clock64() : 0;
#define RQ_STAMP(k) do { if (rq_dbg && tid == 0) atomicAdd((unsigned long long *)&rq_dbg[k], (unsigned long long)(clock64() - rq_t0)); } while (0)
    // boundary lines (cv::LineIterator, 8-connected, left to right), a quarter of the threads per edge.  The iterator's error
    // recurrence  err += -2 minor + (err < 0 ? 2 major : 0)  has the closed form "minor steps taken before pixel i" =
    // round-half-down(minor i / major) = floor((2 minor i + major - 1) / (2 major)), so the pixels of a line are independent.
    {
        const int per = nthreads >> 2, e = tid/per, li = tid - e*per;
        int i0 = (e + 3) & 3, i1 = e;
        long long x1 = s_xy[2*i0], y1 = s_xy[2*i0+1], x2 = s_xy[2*i1], y2 = s_xy[2*i1+1];
        bool ok = e < 4;
        if (ok && ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
                   (unsigned long long)y1 >= (unsigned long long)hh || (unsigned long long)y2 >= (unsigned long long)hh))
            ok = clip_line_dev(w, hh, x1, y1, x2, y2);
        if (ok) {
            long long dx = x2 - x1, dy = y2 - y1;
            if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
            long long sy = dy < 0 ? -1 : 1; if (dy < 0) dy = -dy;
            const bool steep = dy > dx;
            const int major = (int)(steep ? dy : dx), minor = (int)(steep ? dx : dy);
            for (int i = li; i <= major; i += per) {
                const int ci = major > 0 ? (2*minor*i + major - 1)/(2*major) : 0;     // (clipped coordinates: < 2^21)
                const long long x = steep ? x1 + ci : x1 + i, y = steep ? y1 + sy*i : y1 + sy*ci;
                if (x >= 0 && x < w && y >= 0 && y < hh) atomicOr(&mask[(y*w + x) >> 5], 1u << ((y*w + x) & 31));
            }
        }
    }
    RQ_STAMP(0);                                                // (boundary lines)
    // scanline interior (FillEdgeCollection): one thread per row.  Round 6: no array is indexed by a run-time value -- the compacted edge list and the
    // insertion sort of a row's crossings put 64 bytes per lane in SCRATCH (global memory: every xs[k] of the sort a dependent round trip; the quad took 30 k of
    // a mu / sigma workgroup's 55 k cycles, tools/mid_stamps.sh).  The four edges keep their slots with a validity flag, a row's (at most four) crossings are
    // sorted by a five-exchange network with "no crossing" = +infinity: the same spans -- pairs of the sorted crossings, a third one ignored.
    {
        long long ex[4], edx[4]; int ey0[4], ey1[4]; bool ev[4];
        int y_min = 2147483647, y_max = -2147483647, ne = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int i0 = (i + 3) & 3;
            const long long p0x = (long long)s_xy[2*i0]*65536, p0y = s_xy[2*i0+1], p1x = (long long)s_xy[2*i]*65536, p1y = s_xy[2*i+1];   // (x * 2^16: negative x)
            ev[i] = p0y != p1y;
            const bool up = p0y < p1y;
            ey0[i] = (int)(up ? p0y : p1y); ey1[i] = (int)(up ? p1y : p0y); ex[i] = up ? p0x : p1x;
            edx[i] = ev[i] ? div_trunc_small(p1x - p0x, p1y - p0y) : 0;      // (|p1x - p0x| <= 2^32 x 2^16, |p1y - p0y| <= 2^32: int coordinates)
            if (ev[i]) { y_min = min(y_min, ey0[i]); y_max = max(y_max, ey1[i]); ne++; }
        }
        RQ_STAMP(1);                                            // (edge slopes)
        if (ne >= 2 && !(y_max < 0 || y_min >= hh)) {
            if (y_max > hh) y_max = hh;
            const long long INF = 0x7fffffffffffffffLL;
            for (int y = max(y_min, 0) + tid; y < y_max; y += nthreads) {
                long long x0, x1, x2, x3; int na = 0;
                { const bool on = ev[0] && ey0[0] <= y && y < ey1[0]; x0 = on ? ex[0] + (long long)(y - ey0[0])*edx[0] : INF; na += on; }
                { const bool on = ev[1] && ey0[1] <= y && y < ey1[1]; x1 = on ? ex[1] + (long long)(y - ey0[1])*edx[1] : INF; na += on; }
                { const bool on = ev[2] && ey0[2] <= y && y < ey1[2]; x2 = on ? ex[2] + (long long)(y - ey0[2])*edx[2] : INF; na += on; }
                { const bool on = ev[3] && ey0[3] <= y && y < ey1[3]; x3 = on ? ex[3] + (long long)(y - ey0[3])*edx[3] : INF; na += on; }
#define RQ_CX(a_, b_) do { const long long lo_ = a_ < b_ ? a_ : b_, hi_ = a_ < b_ ? b_ : a_; a_ = lo_; b_ = hi_; } while (0)
                RQ_CX(x0, x1); RQ_CX(x2, x3); RQ_CX(x0, x2); RQ_CX(x1, x3); RQ_CX(x1, x2);
#undef RQ_CX
#pragma unroll
                for (int sp = 0; sp < 2; sp++) {
                    if (na < 2*sp + 2) break;
                    const long long xl = sp == 0 ? x0 : x2, xr = sp == 0 ? x1 : x3;
                    int xa = (int)((xl + 65535) >> 16), xb = (int)(xr >> 16);
                    if (xa < w && xb >= 0) { if (xa < 0) xa = 0; if (xb >= w) xb = w - 1;
                        if (xa <= xb) {                                   // the span's bits are contiguous: whole words at a time
                            const int b0 = y*w + xa, b1 = y*w + xb;
                            for (int wd = b0 >> 5; wd <= (b1 >> 5); wd++) {
                                unsigned m = 0xffffffffu;
                                if (wd == (b0 >> 5)) m &= 0xffffffffu << (b0 & 31);
                                if (wd == (b1 >> 5)) m &= 0xffffffffu >> (31 - (b1 & 31));
                                atomicOr(&mask[wd], m);
                            }
                        } }
                }
            }
        }
    }
    RQ_STAMP(2);
#undef RQ_STAMP
}
