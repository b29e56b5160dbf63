/* urf_rdp.c -- see urf_rdp.h.  TEST INFRASTRUCTURE ONLY. */
#include <stdlib.h>
#include <string.h>

#include "urf_rdp.h"

/* squared distance of p to the segment a-b, float arithmetic */
static float seg_dist2(float px, float py, float ax, float ay, float bx, float by)
{
    const float vx = bx - ax, vy = by - ay;
    const float wx = px - ax, wy = py - ay;
    const float c1 = wx * vx + wy * vy;
    if (c1 <= 0.0f)
        return wx * wx + wy * wy;
    const float c2 = vx * vx + vy * vy;
    if (c2 <= c1) {
        const float ux = px - bx, uy = py - by;
        return ux * ux + uy * uy;
    }
    const float b = c1 / c2;
    const float qx = ax + b * vx, qy = ay + b * vy;
    const float dx = px - qx, dy = py - qy;
    return dx * dx + dy * dy;
}

void urf_rdp_float(const float* x, const float* y, int n, float max_distance, unsigned char* keep)
{
    if (n <= 0)
        return;
    if (n < 3 || max_distance < 0.0f) {
        memset(keep, 1, (size_t)n);
        return;
    }
    memset(keep, 0, (size_t)n);
    keep[0] = keep[n - 1] = 1;
    const float md2 = max_distance * max_distance;
    int* stack = (int*)malloc((size_t)n * 2 * sizeof(int));
    int top = 0;
    stack[0] = 0;
    stack[1] = n - 1;
    top = 1;
    while (top > 0) {
        --top;
        const int a = stack[2 * top], b = stack[2 * top + 1];
        if (b - a < 2)
            continue;
        float best = -1.0f;
        int arg = -1;
        for (int i = a + 1; i < b; i++) {
            const float d2 = seg_dist2(x[i], y[i], x[a], y[a], x[b], y[b]);
            if (d2 > best) {
                best = d2;
                arg = i;
            }
        }
        if (arg >= 0 && best > md2) {
            keep[arg] = 1;
            stack[2 * top] = a;
            stack[2 * top + 1] = arg;
            top++;
            stack[2 * top] = arg;
            stack[2 * top + 1] = b;
            top++;
        }
    }
    free(stack);
}
