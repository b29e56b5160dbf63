/*
 * Force-included (-include) into the reference's sources for the oracle variant
 * _ref/urf_ref_libm ONLY.  TEST INFRASTRUCTURE.
 *
 * The reference calls acos / asin / atan2 on float arguments, i.e. glibc's acosf / asinf /
 * atan2f.  Those are not correctly rounded (glibc 2.35: asinf(0.8660254f) is 1 ulp high) and
 * differ between glibc releases; the product defines the three functions by include/urf_libm.h
 * instead ("libm").  This header maps the reference's float calls onto that
 * definition, and nothing else: the reference's control flow, operand types and every other
 * operation stay its own.  It separates "the restatement follows the reference" (oracle B must
 * equal this variant on ANY input, including clouds placed on decision boundaries) from "which
 * libm rounds how" (the plain variant _ref/urf_ref keeps glibc and produces the ordinary goldens).
 */
#pragma once
#include <cmath>
#include <math.h>

#include "urf_libm.h"

inline float urf_shim_acos(float v) { return urf_acosf(v); }
inline float urf_shim_asin(float v) { return urf_asinf(v); }
inline float urf_shim_atan2(float y, float x) { return urf_atan2f(y, x); }
inline double urf_shim_acos(double v) { return ::acos(v); }
inline double urf_shim_asin(double v) { return ::asin(v); }
inline double urf_shim_atan2(double y, double x) { return ::atan2(y, x); }

#define acos(x) urf_shim_acos(x)
#define asin(x) urf_shim_asin(x)
#define atan2(y, x) urf_shim_atan2(y, x)
