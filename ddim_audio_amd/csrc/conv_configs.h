// Tile configurations of conv_mfma_kernel per (dtype, mode, Cin, Cout) of the audio U-Net
// (configs/audio.yml:48: ch=[32,64,96,128,192,256]).  Columns:
//   X(T, CIN, NOUT, NB, MODE, TH, TW, WM, WN, KC, TPC, VAR, OVL)
// VAR 0 = large tile (default), VAR 1 = small tile / split output channels, chosen when the large tiling
// would leave most of the 256 CUs idle (deep levels, small batches).  VAR >= 2 are tuning candidates that only
// run when DDIMX_CONV_VAR=<n> is set (bench A/B); OVL: see ConvCfg::SEPARATE_OUT.
// NOUT is the virtual output width (UP4: 2*Cprev, the two column-parity classes side by side).
// LDS use is static_assert-ed against the 160 KiB budget inside ConvCfg.
#pragma once

#define DDIMX_CONV3_BF16(X)                           \
    X(__bf16, 32, 32, 32, CONV3, 16, 32, 4, 1, 32, 9, 0, 1) \
    X(__bf16, 64, 64, 64, CONV3, 8, 32, 4, 1, 64, 1, 0, 1)  \
    X(__bf16, 96, 96, 96, CONV3, 8, 32, 8, 1, 96, 1, 0, 0)  \
    X(__bf16, 128, 128, 128, CONV3, 8, 16, 4, 1, 128, 1, 0, 0) \
    X(__bf16, 192, 192, 192, CONV3, 8, 16, 4, 1, 64, 1, 0, 0) \
    X(__bf16, 256, 256, 256, CONV3, 16, 8, 4, 1, 32, 1, 0, 0) \
    X(__bf16, 128, 128, 64, CONV3, 8, 8, 2, 2, 128, 1, 1, 0)  \
    X(__bf16, 192, 192, 64, CONV3, 8, 16, 2, 2, 192, 1, 1, 0) \
    X(__bf16, 256, 256, 64, CONV3, 8, 8, 2, 2, 256, 1, 1, 0) \
    X(__bf16, 32, 32, 32, CONV3, 8, 32, 4, 1, 32, 9, 2, 0)  \
    X(__bf16, 32, 32, 32, CONV3, 8, 32, 4, 1, 32, 9, 3, 1)  \
    X(__bf16, 32, 32, 32, CONV3, 16, 32, 8, 1, 32, 9, 4, 0) \
    X(__bf16, 32, 32, 32, CONV3, 16, 32, 8, 1, 32, 9, 5, 1) \
    X(__bf16, 64, 64, 64, CONV3, 8, 32, 4, 1, 64, 9, 2, 1)  \
    X(__bf16, 64, 64, 64, CONV3, 8, 16, 4, 1, 64, 9, 3, 0)  \
    X(__bf16, 64, 64, 64, CONV3, 8, 32, 4, 2, 64, 1, 4, 0)  \
    X(__bf16, 64, 64, 64, CONV3, 8, 32, 4, 2, 64, 9, 5, 1)  \
    X(__bf16, 96, 96, 96, CONV3, 8, 16, 4, 1, 48, 1, 2, 0)  \
    X(__bf16, 96, 96, 96, CONV3, 8, 32, 4, 1, 96, 1, 3, 1)

#define DDIMX_DOWNUP_BF16(X)                           \
    X(__bf16, 32, 64, 64, DOWN4, 8, 16, 4, 2, 32, 16, 0, 0)  \
    X(__bf16, 64, 96, 96, DOWN4, 8, 16, 4, 1, 64, 1, 0, 0)   \
    X(__bf16, 96, 128, 128, DOWN4, 8, 8, 2, 2, 96, 1, 0, 0)  \
    X(__bf16, 128, 192, 192, DOWN4, 8, 8, 2, 2, 32, 1, 0, 0) \
    X(__bf16, 192, 256, 256, DOWN4, 4, 8, 1, 4, 32, 1, 0, 0) \
    X(__bf16, 256, 384, 384, UP4, 8, 8, 2, 2, 32, 1, 0, 0)   \
    X(__bf16, 192, 256, 128, UP4, 8, 16, 4, 2, 96, 1, 0, 0)  \
    X(__bf16, 128, 192, 192, UP4, 8, 8, 2, 2, 64, 1, 0, 0)   \
    X(__bf16, 96, 128, 128, UP4, 8, 16, 4, 1, 96, 1, 0, 0)   \
    X(__bf16, 64, 64, 64, UP4, 16, 32, 8, 1, 64, 6, 0, 0)    \
    X(__bf16, 192, 256, 128, DOWN4, 4, 8, 1, 4, 96, 1, 1, 0)  \
    X(__bf16, 128, 192, 96, DOWN4, 4, 8, 1, 3, 128, 1, 1, 0)  \
    X(__bf16, 256, 384, 128, UP4, 8, 8, 2, 2, 128, 1, 1, 0)

#define DDIMX_CONV3_F32(X)                          \
    X(float, 32, 32, 32, CONV3, 8, 32, 8, 1, 32, 9, 0, 0) \
    X(float, 64, 64, 64, CONV3, 8, 16, 4, 1, 32, 1, 0, 0) \
    X(float, 96, 96, 96, CONV3, 8, 16, 4, 1, 32, 1, 0, 0) \
    X(float, 128, 128, 128, CONV3, 8, 8, 2, 2, 32, 1, 0, 0) \
    X(float, 192, 192, 192, CONV3, 8, 8, 2, 2, 16, 1, 0, 0) \
    X(float, 256, 256, 256, CONV3, 8, 8, 2, 2, 8, 1, 0, 0)

#define DDIMX_DOWNUP_F32(X)                          \
    X(float, 32, 64, 64, DOWN4, 8, 8, 2, 2, 32, 1, 0, 0)   \
    X(float, 64, 96, 96, DOWN4, 8, 8, 2, 3, 32, 1, 0, 0)   \
    X(float, 96, 128, 128, DOWN4, 4, 8, 1, 4, 32, 1, 0, 0) \
    X(float, 128, 192, 192, DOWN4, 4, 8, 1, 3, 16, 1, 0, 0) \
    X(float, 192, 256, 64, DOWN4, 4, 8, 1, 2, 8, 1, 0, 0) \
    X(float, 256, 384, 192, UP4, 4, 8, 1, 3, 16, 1, 0, 0)  \
    X(float, 192, 256, 256, UP4, 4, 8, 1, 4, 16, 1, 0, 0)  \
    X(float, 128, 192, 192, UP4, 4, 8, 1, 3, 32, 1, 0, 0)  \
    X(float, 96, 128, 128, UP4, 8, 8, 2, 2, 32, 1, 0, 0)   \
    X(float, 64, 64, 64, UP4, 8, 16, 4, 1, 64, 1, 0, 0)
