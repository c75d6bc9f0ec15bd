// Micro-benchmark behind DESIGN section 7 "first frames": what a cold handle pays for its queues.
// hipMalloc of large blocks (serial, and from several host threads at once), hipMallocAsync from a pool, the virtual-memory API,
// and whether a kernel running on another stream keeps running while the host allocates.
// build: hipcc --offload-arch=gfx950 -O2 -o alloc_bench alloc_bench.hip -lpthread ; run: ./alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK( x ) do { hipError_t e_ = ( x ); if( e_ != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e_ ) ); return 1; } } while( 0 )
static double now_ms() { return std::chrono::duration< double, std::milli >( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
__global__ void spin( unsigned long long* out, unsigned long long cycles )
{
    unsigned long long t0 = wall_clock64(), t = t0;
    while( t - t0 < cycles ) t = wall_clock64();
    if( threadIdx.x == 0 && blockIdx.x == 0 ) *out = t - t0;
}
__global__ void touch( char* p, size_t n ) { size_t i = ( ( size_t )blockIdx.x * blockDim.x + threadIdx.x ) * 4096; if( i < n ) p[ i ] = 1; }
int main()
{
    CK( hipSetDevice( 0 ) );
    CK( hipFree( 0 ) );
    const size_t GB = ( size_t )1 << 30;
    void* p = nullptr;
    for( int rep = 0; rep < 2; rep++ )
        for( size_t gb : { 1, 4, 8, 16 } )
        {
            double t0 = now_ms(); CK( hipMalloc( &p, gb * GB ) ); double t1 = now_ms();
            hipLaunchKernelGGL( touch, dim3( ( unsigned )( gb * GB / 4096 / 256 ) ), dim3( 256 ), 0, 0, ( char* )p, gb * GB ); CK( hipDeviceSynchronize() ); double t2 = now_ms();
            CK( hipFree( p ) ); double t3 = now_ms();
            printf( "hipMalloc %2zu GB: %7.2f ms (%.2f ms/GB), first touch of every page %7.2f ms, hipFree %7.2f ms\n", gb, t1 - t0, ( t1 - t0 ) / gb, t2 - t1, t3 - t2 );
        }
    // the same 16 GB from 1, 2, 4, 8 host threads at once
    for( int nt : { 1, 2, 4, 8 } )
    {
        std::vector< void* > ptr( nt, nullptr ); std::vector< std::thread > th;
        double t0 = now_ms();
        for( int k = 0; k < nt; k++ ) th.emplace_back( [ &, k ]() { hipSetDevice( 0 ); hipMalloc( &ptr[ k ], 16 * GB / nt ); } );
        for( auto& t : th ) t.join();
        double t1 = now_ms();
        for( int k = 0; k < nt; k++ ) hipFree( ptr[ k ] );
        printf( "16 GB by %d threads at once: %7.2f ms\n", nt, t1 - t0 );
    }
    // many small blocks against one large one (the workspace is 11 blocks per lane)
    { std::vector< void* > ptr( 64 ); double t0 = now_ms(); for( auto& q : ptr ) CK( hipMalloc( &q, 16 * GB / 64 ) ); double t1 = now_ms(); for( auto& q : ptr ) hipFree( q ); printf( "16 GB in 64 blocks: %7.2f ms\n", t1 - t0 ); }
    // stream-ordered pool
    {
        hipStream_t s; CK( hipStreamCreate( &s ) );
        hipMemPool_t pool; CK( hipDeviceGetDefaultMemPool( &pool, 0 ) );
        unsigned long long thr = ~0ull; CK( hipMemPoolSetAttribute( pool, hipMemPoolAttrReleaseThreshold, &thr ) );
        for( int rep = 0; rep < 3; rep++ )
        {
            double t0 = now_ms(); CK( hipMallocAsync( &p, 16 * GB, s ) ); CK( hipStreamSynchronize( s ) ); double t1 = now_ms();
            CK( hipFreeAsync( p, s ) ); CK( hipStreamSynchronize( s ) ); double t2 = now_ms();
            printf( "hipMallocAsync 16 GB (call %d on a pool that keeps its memory): %7.2f ms, hipFreeAsync %7.2f ms\n", rep + 1, t1 - t0, t2 - t1 );
        }
        CK( hipStreamDestroy( s ) );
    }
    // virtual memory API
    {
        hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; hipError_t e = hipMemGetAllocationGranularity( &gran, &prop, hipMemAllocationGranularityRecommended );
        if( e == hipSuccess )
        {
            size_t bytes = 16 * GB; hipMemGenericAllocationHandle_t hnd; void* va = nullptr;
            double t0 = now_ms(); e = hipMemCreate( &hnd, bytes, &prop, 0 ); double t1 = now_ms();
            if( e == hipSuccess ) e = hipMemAddressReserve( &va, bytes, gran, nullptr, 0 );
            if( e == hipSuccess ) e = hipMemMap( va, bytes, 0, hnd, 0 );
            hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if( e == hipSuccess ) e = hipMemSetAccess( va, bytes, &acc, 1 );
            double t2 = now_ms();
            printf( "hipMemCreate 16 GB: %7.2f ms, reserve + map + set access %7.2f ms (%s; granularity %zu)\n", t1 - t0, t2 - t1, hipGetErrorString( e ), gran );
            if( e == hipSuccess ) { hipMemUnmap( va, bytes ); hipMemAddressFree( va, bytes ); hipMemRelease( hnd ); }
        }
        else printf( "virtual memory API: %s\n", hipGetErrorString( e ) );
        ( void )hipGetLastError();
    }
    // does a running kernel notice the allocation?  a ~40 ms spin kernel on a stream, 8 GB allocated by the host meanwhile
    {
        hipStream_t s; CK( hipStreamCreate( &s ) );
        unsigned long long* d; CK( hipMalloc( &d, 8 ) );
        hipEvent_t a, b; CK( hipEventCreate( &a ) ); CK( hipEventCreate( &b ) );
        for( int with = 0; with < 2; with++ )
        {
            CK( hipEventRecord( a, s ) );
            hipLaunchKernelGGL( spin, dim3( 256 ), dim3( 256 ), 0, s, d, 4000000ull );   /* 100 MHz wall clock: 40 ms */
            CK( hipEventRecord( b, s ) );
            double t0 = now_ms(), t1 = t0, t2 = t0;
            if( with ) { CK( hipMalloc( &p, 8 * GB ) ); t1 = now_ms();
                         /* and a launch from this thread right after it, as a lane would */
                         hipLaunchKernelGGL( touch, dim3( 1 ), dim3( 64 ), 0, 0, ( char* )p, 4096 ); t2 = now_ms(); }
            CK( hipEventSynchronize( b ) ); double t3 = now_ms();
            float ms = 0; CK( hipEventElapsedTime( &ms, a, b ) );
            printf( "spin kernel %s: kernel %7.2f ms by events, host: malloc %7.2f ms, next launch %5.2f ms, all done after %7.2f ms\n", with ? "with hipMalloc 8 GB meanwhile" : "alone", ms, t1 - t0, t2 - t1, t3 - t0 );
            if( with ) { CK( hipDeviceSynchronize() ); CK( hipFree( p ) ); }
        }
    }
    // what making a lane costs: streams, events, pinned host memory (render_lanes: ~10 ms per lane)
    {
        hipStream_t st[ 12 ]; hipEvent_t ev[ 30 ]; void* hp[ 6 ]; unsigned long long* d; CK( hipMalloc( &d, 8 ) );
        double t0 = now_ms();
        for( int k = 0; k < 12; k++ ) { double a = now_ms(); CK( hipStreamCreateWithFlags( &st[ k ], hipStreamNonBlocking ) ); printf( "stream %d: %.2f ms\n", k + 1, now_ms() - a ); }
        double t1 = now_ms();
        for( int k = 0; k < 30; k++ ) CK( hipEventCreate( &ev[ k ] ) );
        double t2 = now_ms();
        for( int k = 0; k < 6; k++ ) CK( hipHostMalloc( &hp[ k ], 4096 ) );
        double t3 = now_ms();
        for( int k = 0; k < 12; k++ ) hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, st[ k ], d, 10ull );
        double t4 = now_ms();
        CK( hipDeviceSynchronize() );
        double t5 = now_ms();
        for( int k = 0; k < 12; k++ ) hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, st[ k ], d, 10ull );
        CK( hipDeviceSynchronize() );
        double t6 = now_ms();
        printf( "12 streams %7.2f ms, 30 events %7.2f ms, 6 x hipHostMalloc 4 KB %7.2f ms, first launch on each stream %7.2f ms (+ sync %7.2f), second round %7.2f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5 );
    }
    // ... and whether streams can be made side by side
    for( int nt : { 1, 3, 6 } )
    {
        std::vector< hipStream_t > st( 6 ); std::vector< std::thread > th;
        double t0 = now_ms();
        for( int k = 0; k < nt; k++ ) th.emplace_back( [ &, k ]() { hipSetDevice( 0 ); for( int i = k; i < 6; i += nt ) ( void )hipStreamCreateWithFlags( &st[ i ], hipStreamNonBlocking ); } );
        for( auto& t : th ) t.join();
        double t1 = now_ms();
        printf( "6 more streams by %d threads: %7.2f ms\n", nt, t1 - t0 );
    }
    return 0;
}
