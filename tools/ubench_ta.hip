// What a 64-lane byte gather costs the texture addresser as a function of how the four lanes of a
// quad are spread over cache lines (L1-resident data, many waves per CU, time per wave-instruction).
//   mode 0: all 64 lanes in one 64-B line
//   mode 1: each quad in its own 64-B line (16 lines per instruction)
//   mode 2: each quad split over two 64-B lines that share a 128-B aligned block
//   mode 3: each quad split over two 64-B lines in different 128-B blocks
//   mode 4: each quad split over four 64-B lines (two 128-B blocks)
//   modes 5-10: see the switch
// If mode 2 costs as mode 1, the L1 tags are 128 B wide and a 16x8-voxel line would pay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void gather( const uint8_t* __restrict__ buf, int mode, int iters, unsigned stride, unsigned* out )
{
    const unsigned lane = threadIdx.x & 63u, quad = lane >> 2, q = lane & 3u;
    // 16 KiB window per wave so everything stays in the 32 KiB L1
    const uint8_t* base = buf + ( ( ( blockIdx.x * blockDim.x + threadIdx.x ) >> 6 ) & 1u ) * 16384u;
    unsigned off;
    switch( mode )
    {
    case 0: off = lane; break;
    case 1: off = quad * 256u + q; break;
    case 2: off = quad * 256u + ( q >> 1 ) * 64u + ( q & 1u ); break;
    case 3: off = quad * 256u + ( q >> 1 ) * 128u + ( q & 1u ); break;
    case 4: off = quad * 256u + q * 64u; break;
    // quads inside one line, spread like the raycaster's 2x2 pixel quads over a 8x8-voxel line
    case 5: off = quad * 256u + ( q >> 1 ) * 8u + ( q & 1u ); break;          // 2x2 voxels
    case 6: off = quad * 256u + ( q >> 1 ) * 16u + ( q & 1u ) * 2u; break;   // every other voxel
    // a 16-lane group = 4x4 pixels (Morton) = 4x4 voxels of ONE line; four lines per instruction
    case 7: off = ( lane >> 4 ) * 256u + ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) ) * 8u +
                  ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) ); break;
    // the same 4x4 voxels but the group straddles two lines in x (two quads each side)
    case 8: off = ( lane >> 4 ) * 256u + ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) ) * 8u +
                  ( lane & 1u ) + ( ( lane >> 2 ) & 1u ) * 64u; break;
    // one quad in four split over two lines, the others whole
    case 9: off = quad * 256u + ( ( quad & 3u ) == 0u ? ( q >> 1 ) * 64u + ( q & 1u ) : ( q >> 1 ) * 8u + ( q & 1u ) ); break;
    // strips of a row-major (x fastest, pitch 136) layout
    case 11: off = ( lane >> 3 ) * 64u + ( lane & 7u ); break;              // 8 lines x 8 consecutive bytes
    case 12: off = 20u + lane; break;                                        // 64 consecutive bytes, misaligned
    case 13: off = 20u + lane * 2u; break;                                   // every other byte (2 voxels per pixel)
    case 14: off = 20u + ( lane >> 5 ) * 136u + ( lane & 31u ); break;      // 32x2 strip
    case 15: off = 20u + ( lane >> 4 ) * 136u + ( lane & 15u ); break;      // 16x4 strip
    case 16: off = 20u + ( lane >> 3 ) * 136u + ( lane & 7u ); break;       // 8x8 tile, row-major lanes
    case 17: off = 20u + ( lane >> 5 ) * 136u + ( lane & 31u ) * 2u; break; // 32x2 strip, every other byte
    case 18: off = 20u + ( lane * 3u ) / 2u; break;                          // 1.5 voxels per pixel
    // unmerged lanes (every other byte) over 8, 6 and 3 lines: cost per line
    case 19: off = ( lane >> 3 ) * 64u + ( lane & 7u ) * 2u; break;
    case 20: off = ( lane % 6u ) * 64u + ( lane / 6u ) * 2u; break;
    case 21: off = ( lane % 3u ) * 64u + ( lane / 3u ) * 2u; break;
    // 2x2 pixel quads (Morton) at one voxel per pixel over a tile that straddles 2x2 blocks
    case 22: { const unsigned x = 5u + ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ),
                              y = 3u + ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) );
               off = ( ( y >> 3 ) * 2u + ( x >> 3 ) ) * 512u + ( y & 7u ) * 8u + ( x & 7u ); } break;
    // the same tile with row-major lanes
    case 23: { const unsigned x = 5u + ( lane & 7u ), y = 3u + ( lane >> 3 );
               off = ( ( y >> 3 ) * 2u + ( x >> 3 ) ) * 512u + ( y & 7u ) * 8u + ( x & 7u ); } break;
    // mode 5 with the 16 lines 320 B apart instead of 256 B (do lines 256 B apart share a bank?)
    case 24: off = quad * 320u + ( q >> 1 ) * 8u + ( q & 1u ); break;
    // mode 22 with the four blocks' lines in different 64-B phases (z-slice swizzled by block)
    case 25: { const unsigned x = 5u + ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ),
                              y = 3u + ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) );
               const unsigned b = ( y >> 3 ) * 2u + ( x >> 3 );
               off = b * 512u + b * 64u + ( y & 7u ) * 8u + ( x & 7u ); } break;
    // six lines (tile over 3x2 blocks, 1.5 voxels per pixel), same 64-B phase vs swizzled
    case 26: { const unsigned x = 5u + ( ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ) * 3u ) / 2u,
                              y = 3u + ( ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) ) * 3u ) / 2u;
               const unsigned b = ( y >> 3 ) * 3u + ( x >> 3 );
               off = b * 512u + ( y & 7u ) * 8u + ( x & 7u ); } break;
    case 27: { const unsigned x = 5u + ( ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ) * 3u ) / 2u,
                              y = 3u + ( ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) ) * 3u ) / 2u;
               const unsigned b = ( y >> 3 ) * 3u + ( x >> 3 );
               off = b * 512u + ( ( b * 3u ) & 7u ) * 64u + ( y & 7u ) * 8u + ( x & 7u ); } break;
    // all lanes in one line but as 2x2 quads of an 8x8 block (Morton over the line)
    default: off = ( ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u ) ) * 8u +
                   ( ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u ) ); break;
    }
    unsigned acc = 0, cur = 0;
    for( int i = 0; i < iters; ++i )
    {
        // eight independent gathers per iteration, as the raycast kernel issues them; the window
        // offset walks through four 4-KiB pages (stride is a kernel argument: nothing folds)
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
    const int iters = 2000, blocks = 256 * 8, threads = 256; // 8 workgroups of 4 waves per CU
    for( int mode = 0; mode < 29; ++mode )
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
        printf( "mode %d: %.3f ms -> %.1f ns per wave-gather per CU (%.1f cycles at 2.4 GHz)\n", mode, best,
                best * 1e6 / instrPerCU, best * 1e6 / instrPerCU * 2.4 );
    }
    return 0;
}
