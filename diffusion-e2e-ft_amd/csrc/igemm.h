// igemm.h — parameter block shared by the implicit-GEMM kernel variants.
#pragma once
#include "common.h"

namespace e2eft {

struct IgemmParams {
    const void* x1;
    const void* x2;
    const void* w;
    const void* bias;
    const void* rowadd;
    const void* residual;
    void* out;
    int M, N, K;
    int ldx1, ldx2, c1, cin;
    int hin, win, hl, wl, kh, kw, stride, pad_t, pad_l, hout, wout;
    float up_sh, up_sw;
    int ldw, ldr, ldo;
    int bias_along_m;
    int rows_per_img;
    float alpha;
    int nzi;
    long sa_o, sa_i, sw_o, sw_i, so_o, so_i, sr_o, sr_i;
    int mtiles, ntiles;
};

// v2: 256x128 tile, 8 waves, LDS-DMA (global_load_lds) 3-stage ring — igemm2.hip
int launch_igemm_v2(int dtype, int mode, IgemmParams& p, int nz, hipStream_t s);

}  // namespace e2eft
