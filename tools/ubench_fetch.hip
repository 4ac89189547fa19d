// Calibration of HBM fetch granularity for byte gathers on gfx950 (time-based, no counters):
//   A: every 64-byte line of a 2 GiB buffer is touched once (64 lanes x 1 byte = one line per wave-instruction)
//   B: only the even 64-byte lines are touched (1 GiB touched, same footprint)
//   C: streaming 16 B/lane reference
// If HBM/L2 fill granularity is 128 B, B moves as many bytes as A and takes as long.
// Also prints what rocprofv3 FETCH_SIZE should be compared with (run under --pmc FETCH_SIZE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void touch_lines( const uint8_t* __restrict__ buf, size_t nLines, size_t lineStride, unsigned* out )
{
    const size_t wave = ( (size_t)blockIdx.x * blockDim.x + threadIdx.x ) >> 6;
    const size_t nWaves = ( (size_t)gridDim.x * blockDim.x ) >> 6;
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    // scatter line order so consecutive waves do not hit consecutive lines (like the raycaster)
    for( size_t l = wave; l < nLines; l += nWaves )
    {
        const size_t line = ( l * 2654435761ull ) % nLines;
        acc += buf[line * lineStride + lane];
    }
    if( acc == 0xFFFFFFFFu ) out[0] = acc;
}

__global__ void stream16( const uint4* __restrict__ buf, size_t n, unsigned* out )
{
    unsigned acc = 0;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x )
    {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if( acc == 0x12345678u ) out[0] = acc;
}

int main()
{
    const size_t bytes = 2ull << 30;
    uint8_t* d; unsigned* o;
    hipMalloc( &d, bytes ); hipMalloc( &o, 4 );
    hipMemset( d, 1, bytes );
    hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
    float ms;
    for( int rep = 0; rep < 2; ++rep )
    {
        hipEventRecord( e0 );
        hipLaunchKernelGGL( touch_lines, dim3( 8192 ), dim3( 256 ), 0, 0, d, bytes / 64, (size_t)64, o );
        hipEventRecord( e1 ); hipDeviceSynchronize(); hipEventElapsedTime( &ms, e0, e1 );
        printf( "A all 64B lines   : %.3f ms, %.1f GB/s of touched lines (2 GiB)\n", ms, bytes / ms / 1e6 );
        hipEventRecord( e0 );
        hipLaunchKernelGGL( touch_lines, dim3( 8192 ), dim3( 256 ), 0, 0, d, bytes / 128, (size_t)128, o );
        hipEventRecord( e1 ); hipDeviceSynchronize(); hipEventElapsedTime( &ms, e0, e1 );
        printf( "B even 64B lines  : %.3f ms, %.1f GB/s of touched lines (1 GiB)\n", ms, bytes / 2 / ms / 1e6 );
        hipEventRecord( e0 );
        hipLaunchKernelGGL( stream16, dim3( 8192 ), dim3( 256 ), 0, 0, (const uint4*)d, bytes / 16, o );
        hipEventRecord( e1 ); hipDeviceSynchronize(); hipEventElapsedTime( &ms, e0, e1 );
        printf( "C stream 16B/lane : %.3f ms, %.1f GB/s (2 GiB)\n", ms, bytes / ms / 1e6 );
    }
    return 0;
}
