/* acn_detmath.h -- deterministic IEEE-binary64 elementary functions shared by host C and HIP device code.
 *
 * Why: the reference seeds the Monte-Carlo stream of every shading point from the bit pattern of the hit
 * position and normal (src/vectors.h:177-190, src/scene.c:537), so a 1-ulp difference in sin/cos/acos/tan/log
 * anywhere upstream yields an unrelated sample stream and a pixel that differs at Monte-Carlo noise level
 * (mean |delta| 0.037 from toggling FMA contraction alone).  The reference calls libm
 * (sqrt acos sin cos tan pow log frexp); libm does not exist on the GPU, and glibc / ocml round differently.
 * These routines use only + - * / sqrt, integer ops and comparisons in a fixed order, so any IEEE-754
 * machine compiled WITHOUT floating-point contraction (-ffp-contract=off) returns identical bits.
 *
 * Accuracy (tests/test_detmath.py, vs glibc): sin, cos, acos, log, exp <= 1 ulp; tan <= 2 ulp; pow(x,y) is
 * exp(y*log x), relative error <= |y log x| * 2^-52 (it only scales colours: src/vectors.h:372, src/scene.c:656).
 *
 * The polynomial/rational kernels follow the classic fdlibm formulations (e_acos.c, e_log.c, e_exp.c,
 * k_sin.c, k_cos.c, e_rem_pio2.c):
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunSoft, a Sun
 *   Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this software is freely
 *   granted, provided that this notice is preserved.
 */
#ifndef ACN_DETMATH_H
#define ACN_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define ACN_HD __host__ __device__ static inline
#else
#define ACN_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

ACN_HD uint64_t acn_f64_bits( double x )
{
    union { double d; uint64_t u; } v; v.d = x; return v.u;
}

ACN_HD double acn_bits_f64( uint64_t u )
{
    union { double d; uint64_t u; } v; v.u = u; return v.d;
}

ACN_HD double acn_fabs( double x ) { return acn_bits_f64( acn_f64_bits( x ) & 0x7FFFFFFFFFFFFFFFull ); }

/* correctly rounded on x86-64 (sqrtsd) and on gfx950 (checked bit-for-bit by tests/test_gpu_detmath.py) */
ACN_HD double acn_sqrt( double x ) { return __builtin_sqrt( x ); }

/* 2^k for -1022 <= k <= 1023 */
ACN_HD double acn_pow2i( int k ) { return acn_bits_f64( ( uint64_t )( k + 1023 ) << 52 ); }

/* mantissa of frexp(): x = m * 2^e, 0.5 <= |m| < 1; 0, inf, nan returned unchanged (C99 frexp) */
ACN_HD double acn_frexp_mant( double x )
{
    uint64_t u = acn_f64_bits( x );
    int e = ( int )( ( u >> 52 ) & 0x7FF );
    if( e == 0x7FF ) return x;
    if( e == 0 )
    {
        if( ( u << 1 ) == 0 ) return x;
        /* subnormal: scale by 2^54 (exact) */
        x = x * 18014398509481984.0;
        u = acn_f64_bits( x );
    }
    u = ( u & 0x800FFFFFFFFFFFFFull ) | 0x3FE0000000000000ull;
    return acn_bits_f64( u );
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* argument reduction x = n*pi/2 + (y0 + y1), |y0| <= ~pi/4; valid for |x| < 2^20*pi/2 (callers pass |x| <= ~2pi) */
ACN_HD int acn_rem_pio2( double x, double* y0, double* y1 )
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1  = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_2  = 6.07710050630396597660e-11; /* second 33 bits */
    const double pio2_2t = 2.02226624879595063154e-21; /* pi/2 - (pio2_1+pio2_2) */
    double fn = ( double )( int )( x * invpio2 + 0.5 );  /* x >= 0 */
    double t  = x - fn * pio2_1;   /* exact: fn*pio2_1 has <= 53 bits */
    double w  = fn * pio2_2;
    double r  = t - w;
    w  = fn * pio2_2t - ( ( t - r ) - w );
    *y0 = r - w;
    *y1 = ( r - *y0 ) - w;
    return ( int )fn;
}

/* sin on [-pi/4, pi/4] with tail y */
ACN_HD double acn_k_sin( double x, double y )
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * ( S3 + z * ( S4 + z * ( S5 + z * S6 ) ) );
    return x - ( ( z * ( 0.5 * y - v * r ) - y ) - v * S1 );
}

/* cos on [-pi/4, pi/4] with tail y */
ACN_HD double acn_k_cos( double x, double y )
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z  = x * x;
    double w  = z * z;
    double r  = z * ( C1 + z * ( C2 + z * C3 ) ) + w * w * ( C4 + z * ( C5 + z * C6 ) );
    double hz = 0.5 * z;
    double a  = 1.0 - hz;
    return a + ( ( ( 1.0 - a ) - hz ) + ( z * r - x * y ) );
}

ACN_HD void acn_sincos( double x, double* s, double* c )
{
    double ax = acn_fabs( x );
    double y0, y1;
    int n = acn_rem_pio2( ax, &y0, &y1 );
    double ks = acn_k_sin( y0, y1 );
    double kc = acn_k_cos( y0, y1 );
    double ss, cc;
    switch( n & 3 )
    {
        case 0:  ss =  ks; cc =  kc; break;
        case 1:  ss =  kc; cc = -ks; break;
        case 2:  ss = -ks; cc = -kc; break;
        default: ss = -kc; cc =  ks; break;
    }
    *s = ( x < 0 ) ? -ss : ss;
    *c = cc;
}

ACN_HD double acn_sin( double x ) { double s, c; acn_sincos( x, &s, &c ); return s; }
ACN_HD double acn_cos( double x ) { double s, c; acn_sincos( x, &s, &c ); return c; }
ACN_HD double acn_tan( double x ) { double s, c; acn_sincos( x, &s, &c ); return s / c; }

/* ------------------------------------------------------------------------------------------------------------------ */
ACN_HD double acn_acos( double x )
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double pS0 =  1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 =  2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 =  7.91534994289814532176e-04, pS5 =  3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 =  2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 =  7.70381505559019352791e-02;
    double ax = acn_fabs( x );
    if( !( ax < 1.0 ) )
    {
        if( x == 1.0 ) return 0.0;
        if( x == -1.0 ) return pi + 2.0 * pio2_lo;
        return ( x - x ) / ( x - x ); /* nan */
    }
    /* The three ranges of e_acos.c share one evaluation of the rational term: z is chosen per range, p( z ) / q( z ) and --
     * for the two outer ranges -- sqrt( z ) are computed once.  Per argument the operations and their order are those of
     * the three separate branches (identical bits); a wavefront whose lanes fall into different ranges (the cosines of
     * the Oren-Nayar term do) runs the 11 multiply-adds and the division once instead of up to three times. */
    const int small = ax < 0.5;
    if( small && ax < 6.938893903907228e-18 ) return pio2_hi + pio2_lo; /* 2^-57 */
    double z = small ? x * x : ( x < 0 ? ( 1.0 + x ) * 0.5 : ( 1.0 - x ) * 0.5 );
    double p = z * ( pS0 + z * ( pS1 + z * ( pS2 + z * ( pS3 + z * ( pS4 + z * pS5 ) ) ) ) );
    double q = 1.0 + z * ( qS1 + z * ( qS2 + z * ( qS3 + z * qS4 ) ) );
    double r = p / q;
    if( small ) return pio2_hi - ( x - ( pio2_lo - x * r ) );
    double s = acn_sqrt( z );
    if( x < 0 )
    {
        double w = r * s - pio2_lo;
        return pi - 2.0 * ( s + w );
    }
    {
        double df = acn_bits_f64( acn_f64_bits( s ) & 0xFFFFFFFF00000000ull );
        double c  = ( z - df * df ) / ( s + df );
        double w  = r * s + c;
        return 2.0 * ( df + w );
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
ACN_HD double acn_log( double x )
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = acn_f64_bits( x );
    uint32_t hx = ( uint32_t )( u >> 32 );
    int k = 0;
    if( hx < 0x00100000u || ( hx >> 31 ) )
    {
        if( ( u << 1 ) == 0 ) return -1.0 / ( x * x );       /* log(+-0) = -inf */
        if( hx >> 31 ) return ( x - x ) / 0.0;               /* log(-#) = nan */
        k -= 54;                                             /* subnormal: scale up */
        x *= 18014398509481984.0;
        u = acn_f64_bits( x );
        hx = ( uint32_t )( u >> 32 );
    }
    else if( hx >= 0x7FF00000u )
    {
        return x;
    }
    else if( hx == 0x3FF00000u && ( u << 32 ) == 0 )
    {
        return 0.0;
    }
    /* reduce x into [sqrt(2)/2, sqrt(2)] */
    hx += 0x3FF00000u - 0x3FE6A09Eu;
    k  += ( int )( hx >> 20 ) - 0x3FF;
    hx  = ( hx & 0x000FFFFFu ) + 0x3FE6A09Eu;
    u   = ( ( uint64_t )hx << 32 ) | ( u & 0xFFFFFFFFull );
    x   = acn_bits_f64( u );

    double f    = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s    = f / ( 2.0 + f );
    double z    = s * s;
    double w    = z * z;
    double t1   = w * ( Lg2 + w * ( Lg4 + w * Lg6 ) );
    double t2   = z * ( Lg1 + w * ( Lg3 + w * ( Lg5 + w * Lg7 ) ) );
    double R    = t2 + t1;
    double dk   = ( double )k;
    return s * ( hfsq + R ) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* ------------------------------------------------------------------------------------------------------------------ */
ACN_HD double acn_exp( double x )
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if( x != x ) return x;
    if( x > 709.782712893383973096 ) return acn_bits_f64( 0x7FF0000000000000ull );
    if( x < -745.13321910194110842 ) return 0.0;
    double ax = acn_fabs( x );
    double hi, lo;
    int k;
    if( ax > 0.34657359027997264 ) /* 0.5 ln2 */
    {
        k  = ( int )( invln2 * x + ( x < 0 ? -0.5 : 0.5 ) );
        hi = x - ( double )k * ln2HI;
        lo = ( double )k * ln2LO;
        x  = hi - lo;
    }
    else if( ax > 3.725290298461914e-09 ) /* 2^-28 */
    {
        k = 0; hi = x; lo = 0;
    }
    else
    {
        return 1.0 + x;
    }
    double xx = x * x;
    double c  = x - xx * ( P1 + xx * ( P2 + xx * ( P3 + xx * ( P4 + xx * P5 ) ) ) );
    double y  = 1.0 + ( x * c / ( 2.0 - c ) - lo + hi );
    if( k == 0 ) return y;
    /* scalbn(y,k) in two exact steps so that subnormal results round once */
    if( k > 1000 )  return y * acn_pow2i( 1000 ) * acn_pow2i( k - 1000 );
    if( k < -1000 ) return y * acn_pow2i( -1000 ) * acn_pow2i( k + 1000 );
    return y * acn_pow2i( k );
}

/* pow for x >= 0 (colours); x < 0 -> nan like libm for non-integer y */
ACN_HD double acn_pow( double x, double y )
{
    if( y == 1.0 ) return x;
    if( y == 0.0 ) return 1.0;
    if( x == 1.0 ) return 1.0;
    if( x != x || y != y ) return x + y;
    if( x < 0 ) return ( x - x ) / ( x - x );
    if( x == 0.0 ) return ( y > 0 ) ? 0.0 : acn_bits_f64( 0x7FF0000000000000ull );
    return acn_exp( y * acn_log( x ) );
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* asin / atan / atan2: only the texture projection of a sphere uses them (src/objects.c:602-617); same fdlibm forms */
ACN_HD double acn_asin( double x )
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
    const double pS0 =  1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 =  2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 =  7.91534994289814532176e-04, pS5 =  3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 =  2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 =  7.70381505559019352791e-02;
    double ax = acn_fabs( x );
    if( !( ax < 1.0 ) )
    {
        if( ax == 1.0 ) return x * pio2_hi + x * pio2_lo;
        return ( x - x ) / ( x - x );
    }
    if( ax < 0.5 )
    {
        if( ax < 7.450580596923828e-09 ) return x; /* 2^-27 */
        double t = x * x;
        double p = t * ( pS0 + t * ( pS1 + t * ( pS2 + t * ( pS3 + t * ( pS4 + t * pS5 ) ) ) ) );
        double q = 1.0 + t * ( qS1 + t * ( qS2 + t * ( qS3 + t * qS4 ) ) );
        return x + x * ( p / q );
    }
    double w = 1.0 - ax;
    double t = w * 0.5;
    double p = t * ( pS0 + t * ( pS1 + t * ( pS2 + t * ( pS3 + t * ( pS4 + t * pS5 ) ) ) ) );
    double q = 1.0 + t * ( qS1 + t * ( qS2 + t * ( qS3 + t * qS4 ) ) );
    double s = acn_sqrt( t );
    double r;
    if( ax >= 0.975 )
    {
        w = p / q;
        r = pio2_hi - ( 2.0 * ( s + s * w ) - pio2_lo );
    }
    else
    {
        double ws = acn_bits_f64( acn_f64_bits( s ) & 0xFFFFFFFF00000000ull );
        double c = ( t - ws * ws ) / ( s + ws );
        double rr = p / q;
        double pp = 2.0 * s * rr - ( pio2_lo - 2.0 * c );
        double qq = pio4_hi - 2.0 * ws;
        r = pio4_hi - ( pp - qq );
    }
    return ( x > 0 ) ? r : -r;
}

ACN_HD double acn_atan( double x )
{
    const double atanhi[ 4 ] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    const double atanlo[ 4 ] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    const double aT0 =  3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 =  1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 =  9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 =  6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 =  4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    if( x != x ) return x;
    double ax = acn_fabs( x );
    int id;
    double hi = 0, lo = 0;
    if( ax >= 7.378697629483821e+19 )   /* 2^66 */
    {
        double r = atanhi[ 3 ] + atanlo[ 3 ];
        return ( x < 0 ) ? -r : r;
    }
    if( ax < 0.4375 )
    {
        if( ax < 1.862645149230957e-09 ) return x; /* 2^-29 */
        id = -1;
        ax = ax;
    }
    else if( ax < 1.1875 )
    {
        if( ax < 0.6875 ) { id = 0; ax = ( 2.0 * ax - 1.0 ) / ( 2.0 + ax ); }
        else              { id = 1; ax = ( ax - 1.0 ) / ( ax + 1.0 ); }
    }
    else
    {
        if( ax < 2.4375 ) { id = 2; ax = ( ax - 1.5 ) / ( 1.0 + 1.5 * ax ); }
        else              { id = 3; ax = -1.0 / ax; }
    }
    if( id >= 0 ) { hi = atanhi[ id ]; lo = atanlo[ id ]; }
    double z = ax * ax;
    double w = z * z;
    double s1 = z * ( aT0 + w * ( aT2 + w * ( aT4 + w * ( aT6 + w * ( aT8 + w * aT10 ) ) ) ) );
    double s2 = w * ( aT1 + w * ( aT3 + w * ( aT5 + w * ( aT7 + w * aT9 ) ) ) );
    double r;
    if( id < 0 ) r = ax - ax * ( s1 + s2 );
    else         r = hi - ( ( ax * ( s1 + s2 ) - lo ) - ax );
    return ( x < 0 ) ? -r : r;
}

ACN_HD double acn_atan2( double y, double x )
{
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    if( x != x || y != y ) return x + y;
    if( x == 1.0 ) return acn_atan( y );
    int sy = ( acn_f64_bits( y ) >> 63 ) != 0;
    int sx = ( acn_f64_bits( x ) >> 63 ) != 0;
    if( y == 0.0 ) return sx ? ( sy ? -pi : pi ) : y;
    if( x == 0.0 ) return sy ? -pi * 0.5 : pi * 0.5;
    double ax = acn_fabs( x ), ay = acn_fabs( y );
    const double inf = acn_bits_f64( 0x7FF0000000000000ull );
    if( ax == inf )
    {
        if( ay == inf ) { double r = sx ? 3.0 * pi * 0.25 : pi * 0.25; return sy ? -r : r; }
        double r = sx ? pi : 0.0;
        return sy ? -r : r;
    }
    if( ay == inf ) return sy ? -pi * 0.5 : pi * 0.5;
    double z;
    double q = ay / ax;
    if( q > 1.8446744073709552e+19 )        z = pi * 0.5 + 0.5 * pi_lo;   /* |y/x| > 2^64 */
    else if( sx && q < 5.421010862427522e-20 ) z = 0.0;                    /* |y/x| < 2^-64, x < 0 */
    else                                    z = acn_atan( q );
    if( !sx ) return sy ? -z : z;
    double r = pi - ( z - pi_lo );
    return sy ? -r : r;
}

/* llrint(): round to nearest even (the default rounding mode), then convert */
ACN_HD long long acn_llrint( double x ) { return ( long long )__builtin_rint( x ); }

#endif /* ACN_DETMATH_H */
