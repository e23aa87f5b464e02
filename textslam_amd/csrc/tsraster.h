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
#define MS_MASK_WORDS 9600        /* 640*480/32 bits */
// cv::fillPoly of one quad (integer corners s_xy, image w x hh) into an LDS bit mask: boundary lines with cv::LineIterator
// (8-connected) by 4 threads, interior by FillEdgeCollection scanlines (16.16 fixed point), one thread per row.  The caller
// clears the mask and synchronises before and after.
__device__ void raster_quad(unsigned *mask, const int *s_xy, int w, int hh, int tid, int nthreads) {
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
    // scanline interior (FillEdgeCollection): one thread per row
    {
        long long ex[4], edx[4]; int ey0[4], ey1[4], ne = 0;
        int y_min = 2147483647, y_max = -2147483647;
        for (int i = 0; i < 4; i++) {
            int i0 = (i + 3) & 3;
            long long p0x = (long long)s_xy[2*i0]*65536, p0y = s_xy[2*i0+1], p1x = (long long)s_xy[2*i]*65536, p1y = s_xy[2*i+1];   // (x * 2^16: negative x)
            if (p0y == p1y) continue;
            if (p0y < p1y) { ey0[ne] = (int)p0y; ey1[ne] = (int)p1y; ex[ne] = p0x; } else { ey0[ne] = (int)p1y; ey1[ne] = (int)p0y; ex[ne] = p1x; }
            edx[ne] = (p1x - p0x)/(p1y - p0y);
            y_min = min(y_min, ey0[ne]); y_max = max(y_max, ey1[ne]); ne++;
        }
        if (ne >= 2 && !(y_max < 0 || y_min >= hh)) {
            if (y_max > hh) y_max = hh;
            for (int y = max(y_min, 0) + tid; y < y_max; y += nthreads) {
                long long xs[4]; int na = 0;
                for (int i = 0; i < ne; i++) if (ey0[i] <= y && y < ey1[i]) xs[na++] = ex[i] + (long long)(y - ey0[i])*edx[i];
                for (int i = 1; i < na; i++) { long long v = xs[i]; int k = i - 1; while (k >= 0 && xs[k] > v) { xs[k+1] = xs[k]; k--; } xs[k+1] = v; }
                for (int i = 0; i + 1 < na; i += 2) {
                    int xa = (int)((xs[i] + 65535) >> 16), xb = (int)(xs[i+1] >> 16);
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
}
