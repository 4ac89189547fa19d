// Developer check: byte-aligned 16-bit and 2-byte-aligned 32-bit global loads (what a packed struct access compiles
// to on amdhsa: one global_load_ushort / global_load_dword) return the bytes at that address on gfx950, and what a
// 64-lane gather of such pairs costs next to two byte gathers.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench_unaligned tools/ubench_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
struct __attribute__((packed)) u16p { uint16_t v; };
struct __attribute__((packed)) u32p { uint32_t v; };
__global__ void check( const uint8_t* __restrict__ a, const uint16_t* __restrict__ b, uint32_t n, uint32_t* bad )
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i + 2 >= n ) return;
    const uint32_t p8 = reinterpret_cast< const u16p* >( a + i )->v;
    if( p8 != ( (uint32_t)a[i] | ( (uint32_t)a[i + 1] << 8 ) ) ) atomicAdd( bad, 1u );
    const uint32_t p16 = reinterpret_cast< const u32p* >( b + i )->v;
    if( p16 != ( (uint32_t)b[i] | ( (uint32_t)b[i + 1] << 16 ) ) ) atomicAdd( bad + 1, 1u );
}
template < int MODE > __global__ void gather( const uint8_t* __restrict__ a, int iters, uint32_t* out )
{
    // an 8x8 tile over micro-block rows, one voxel per pixel, as the trilinear taps of a z-slice
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t off = ( blockIdx.x * 4096u ) + ( lane >> 3 ) * 8u + ( lane & 7u ), acc = 0;
    for( int i = 0; i < iters; ++i )
    {
        if( MODE == 0 ) acc += (uint32_t)a[off] + (uint32_t)a[off + 1u];
        else acc += reinterpret_cast< const u16p* >( a + off )->v;
        off = ( off + 64u + ( acc & 1u ) ) & 0x1FFFFFu; /* stays in the first half: off + 1 is always inside the buffer */
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const uint32_t n = 1u << 22;
    std::vector< uint8_t > h( n ); std::vector< uint16_t > h16( n );
    for( uint32_t i = 0; i < n; ++i ) { h[i] = (uint8_t)( i * 2654435761u >> 24 ); h16[i] = (uint16_t)( i * 2246822519u >> 16 ); }
    uint8_t* a; uint16_t* b; uint32_t* bad; uint32_t* out;
    hipMalloc( &a, n ); hipMalloc( &b, n * 2 ); hipMalloc( &bad, 8 ); hipMalloc( &out, 1024 * 256 * 4 );
    hipMemcpy( a, h.data(), n, hipMemcpyHostToDevice ); hipMemcpy( b, h16.data(), n * 2, hipMemcpyHostToDevice ); hipMemset( bad, 0, 8 );
    hipLaunchKernelGGL( check, dim3( n / 256 ), dim3( 256 ), 0, 0, a, b, n, bad );
    uint32_t hb[2]; hipMemcpy( hb, bad, 8, hipMemcpyDeviceToHost );
    printf( "unaligned u8-pair mismatches %u, u16-pair mismatches %u (of %u addresses, every alignment)\n", hb[0], hb[1], n );
    for( int mode = 0; mode < 2; ++mode )
    {
        hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
        float best = 1e9f;
        for( int r = 0; r < 5; ++r )
        {
            hipEventRecord( e0 );
            if( mode == 0 ) hipLaunchKernelGGL( gather< 0 >, dim3( 1024 ), dim3( 256 ), 0, 0, a, 2000, out );
            else hipLaunchKernelGGL( gather< 1 >, dim3( 1024 ), dim3( 256 ), 0, 0, a, 2000, out );
            hipEventRecord( e1 ); hipEventSynchronize( e1 );
            float ms; hipEventElapsedTime( &ms, e0, e1 ); best = ms < best ? ms : best;
        }
        printf( "%s: %.3f ms for 1024 x 4 waves x 2000 steps\n", mode == 0 ? "two byte gathers" : "one pair gather  ", best );
    }
    return hb[0] + hb[1] ? 1 : 0;
}
