// Internal declarations shared by sr_host.cpp (pure host bookkeeping) and sr_engine.hip.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

#include "sr_hip.h"

#define SR_MAX_LEVELS 16

int sr_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

struct SrWin {
    int a, b;  // [a, b)
    bool empty() const { return a >= b; }
};

// Row windows of one tile for a canvas row range (see DESIGN.md "strip windows").
struct SrTileLevels {
    int nl;                           // levels actually built (>= 1)
    int H[SR_MAX_LEVELS], W[SR_MAX_LEVELS];
    SrWin gw[SR_MAX_LEVELS];          // rows of G_i (i == 0: rows of the input tile) that are read
    SrWin rw[SR_MAX_LEVELS];          // rows of R_i that are produced (i >= 1)
    SrWin cw;                         // tile-local rows the final gather reads
};

void sr_level_dims(int h, int w, int levels, int *nl, int *H, int *W);
void sr_plan_windows(int tile_h, int tile_w, int tile_y, int levels, int row_begin, int row_end,
                     int canvas_h, SrTileLevels *out);
