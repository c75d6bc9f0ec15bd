/* test shim: evaluates actinon_amd/csrc/acn_detmath.h on the host CPU over arrays (same op codes as acn_detmath_eval) */
#include <stddef.h>
#include "acn_detmath.h"

void detmath_eval( int op, const double* x, const double* y, double* out, size_t n )
{
    for( size_t i = 0; i < n; i++ )
    {
        double a = x[ i ], b = y ? y[ i ] : 0.0, r = 0;
        switch( op )
        {
            case 0: r = acn_sin( a ); break;
            case 1: r = acn_cos( a ); break;
            case 2: r = acn_tan( a ); break;
            case 3: r = acn_acos( a ); break;
            case 4: r = acn_log( a ); break;
            case 5: r = acn_exp( a ); break;
            case 6: r = acn_pow( a, b ); break;
            case 7: r = acn_sqrt( a ); break;
            case 8: r = a / b; break;
            case 9: r = ( double )acn_f64_bits( a ); break;
            case 10: r = acn_frexp_mant( a ); break;
            default: break;
        }
        out[ i ] = r;
    }
}
