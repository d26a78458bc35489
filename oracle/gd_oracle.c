/*
 * gd_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracle for the Gaussian-distance losses: a from-scratch scalar restatement of
 * /root/reference/mmdet3d_gaussian/models/losses/gaussian_distance_loss.py:8-310
 * (function-by-function citations in gd_oracle_body.inc).  Built twice: fp64 (the
 * arbiter the HIP kernels are graded against) and fp32 (the timed `cpu_baseline` "port"
 * in bench.py).
 *
 * Pinning: tests/test_oracle_gd.py checks both builds against tests/golden/gd_pairs.npz /
 * gd_module.npz, which were produced by importing and running the REAL reference module in
 * the build container (tests/golden/make_golden_gd.py).  Parity is therefore PINNED for the
 * loss arithmetic; the only non-reference arithmetic in those fixtures is the stub of mmdet's
 * `weighted_loss` reduction (mmdet is absent from the image).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (mmdet3d-gaussian_amd/) never does and has no CPU fallback.
 */
#include <math.h>
#include <stdint.h>

#include "../include/gd3d.h"

#define REAL double
#define SFX _f64
#define COS cos
#define SIN sin
#define LOG log
#define EXP exp
#define SQRT sqrt
#define LOG1P log1p
#define EXPM1 expm1
#include "gd_oracle_body.inc"
#undef REAL
#undef SFX
#undef COS
#undef SIN
#undef LOG
#undef EXP
#undef SQRT
#undef LOG1P
#undef EXPM1

#define REAL float
#define SFX _f32
#define COS cosf
#define SIN sinf
#define LOG logf
#define EXP expf
#define SQRT sqrtf
#define LOG1P log1pf
#define EXPM1 expm1f
#include "gd_oracle_body.inc"
