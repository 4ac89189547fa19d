// What a PARTIALLY ACTIVE 64-lane byte gather costs the texture addresser: the question behind ray compaction
// after early ray termination (north star: "wave-level ballot for ERT compaction").  L1-resident data, many
// waves per CU, time per wave-instruction.  Active lanes read like the raycaster's tile (2x2-pixel quads over
// 8x8-voxel lines, 1.5 voxels per pixel, tile over 3x2 micro-blocks); the others are switched off by EXEC.
//   mode 0: all 64 lanes
//   mode 1: 32 lanes: the upper half of the tile (8 whole quads)            -- survivors stay together
//   mode 2: 32 lanes: two lanes of every quad (16 half quads)               -- survivors scattered
//   mode 3: 16 lanes: one quarter of the tile (4 whole quads)
//   mode 4: 16 lanes: one lane of every quad
//   mode 5: 16 lanes scattered as in mode 4, then COMPACTED into lanes 0-15 (4 quads whose lanes are pixels
//           from all over the tile: what a ballot + prefix-sum compaction inside the wave would produce)
//   mode 6: 8 lanes: one lane of every other quad          mode 7: the same compacted into lanes 0-7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ unsigned tile_offset( unsigned lane )
{
    const unsigned x = 5u + ( ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ) * 3u ) / 2u,
                   y = 3u + ( ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) ) * 3u ) / 2u;
    const unsigned b = ( y >> 3 ) * 3u + ( x >> 3 );
    return b * 512u + ( y & 7u ) * 8u + ( x & 7u );
}

__global__ void gather( const uint8_t* __restrict__ buf, int mode, int iters, unsigned stride, unsigned* out )
{
    const unsigned lane = threadIdx.x & 63u;
    const uint8_t* base = buf + ( ( ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6 ) & 1u ) * 16384u;
    bool active = true;
    unsigned off = tile_offset( lane );
    switch( mode )
    {
    case 1: active = lane >= 32u; break;
    case 2: active = ( lane & 2u ) == 0u; break;
    case 3: active = lane >= 48u; break;
    case 4: active = ( lane & 3u ) == 0u; break;
    case 5: active = lane < 16u; off = tile_offset( lane * 4u ); break;
    case 6: active = ( lane & 7u ) == 0u; break;
    case 7: active = lane < 8u; off = tile_offset( lane * 8u ); break;
    // which halves of a 2x2-pixel quad are cheap?  (lane & 1 = x, lane & 2 = y inside the quad)
    case 8: active = ( lane & 2u ) != 0u; break;                 // lanes 2,3: the lower pixel row of every quad
    case 9: active = ( lane & 1u ) == 0u; break;                 // lanes 0,2: the left pixel column of every quad
    case 10: active = ( ( lane ^ ( lane >> 1 ) ) & 1u ) == 0u; break; // lanes 0,3: the diagonal
    case 11: active = ( lane & 3u ) != 3u; break;                // three lanes of every quad
    // two instructions' worth in one: every quad complete, but only every other quad
    case 12: active = ( lane & 4u ) == 0u; break;                // quads 0,2,4,...: 8 whole quads spread over the tile
    default: break;
    }
    unsigned acc = 0, cur = 0;
    if( active )
        for( int i = 0; i < iters; ++i )
        {
            unsigned v[8];
#pragma unroll
            for( int k = 0; k < 8; ++k )
            {
                v[k] = base[off + cur];
                cur = ( cur + stride ) & 12288u;
            }
#pragma unroll
            for( int k = 0; k < 8; ++k )
                acc += v[k];
            asm volatile( "" : "+v"( acc ) );
        }
    if( acc == 0xFFFFFFFFu ) out[0] = acc;
}

int main()
{
    uint8_t* d; unsigned* o;
    hipMalloc( &d, 1 << 20 ); hipMalloc( &o, 4 );
    hipMemset( d, 1, 1 << 20 );
    hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
    const int iters = 2000, blocks = 256 * 8, threads = 256;
    const int lanes[13] = { 64, 32, 32, 16, 16, 16, 8, 8, 32, 32, 32, 48, 32 };
    for( int mode = 0; mode < 13; ++mode )
    {
        float best = 1e9f;
        for( int rep = 0; rep < 3; ++rep )
        {
            float ms;
            hipEventRecord( e0 );
            hipLaunchKernelGGL( gather, dim3( blocks ), dim3( threads ), 0, 0, d, mode, iters, 4096u, o );
            hipEventRecord( e1 ); hipDeviceSynchronize(); hipEventElapsedTime( &ms, e0, e1 );
            best = ms < best ? ms : best;
        }
        const double instrPerCU = (double)blocks * ( threads / 64 ) * iters * 8 / 256.0;
        printf( "mode %d (%2d active lanes): %.3f ms -> %.1f cycles per wave-gather at 2.4 GHz, %.2f per active lane\n", mode,
                lanes[mode], best, best * 1e6 / instrPerCU * 2.4, best * 1e6 / instrPerCU * 2.4 / lanes[mode] );
    }
    return 0;
}
