/*
 * vrc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the volume raycaster.
 *
 * The per-ray hot path of renderers/cudaRaycaster/cuda/Renderer.cu:95-230 re-designed for
 * CDNA4:
 *   - one wave64 = one 8x8 pixel tile, one workgroup = one wave, so the hardware
 *     dispatcher load-balances tiles (ray lengths differ by >2x across the image) and
 *     no barrier is needed after the table is staged;
 *   - workgroup -> tile mapping is XCD-aware: the 8 XCDs each take a contiguous band of
 *     tile rows, so neighbouring tiles (which share atlas micro-blocks) hit the same L2;
 *   - the atlas is read as 8x8x8-voxel micro-blocks (one z-slice of a block = one 64-byte
 *     segment), so the 64 fetches of a wave step land in a handful of cache lines;
 *   - TF lookup + opacity correction are folded into a 256-entry classified table staged
 *     in LDS once per workgroup (4 KiB): no per-sample pow, one ds_read_b128 per sample;
 *   - bricks are enumerated by a DDA over the brick grid instead of the O(nodes) loop;
 *   - no MFMA: this is byte gather + scalar compositing, there is no contraction.
 */
#include "vrc_internal.h"

#define VRC_TILE 8u
#define VRC_WG 64u

/* classified-sample table (vrc_core.h: vrc_lut_entry): 256 entries per frame instead of a TF
 * fetch + pow per sample */
__global__ void vrc_k_build_lut( const float* __restrict__ tf, vrc_f4* __restrict__ lut,
                                 vrc_lut_params p )
{
    const uint32_t d = threadIdx.x;
    if( d < 256u )
        lut[d] = vrc_lut_entry( tf, d, p );
}

hipError_t vrc_launch_build_lut( const float* tf, vrc_f4* lut, vrc_lut_params p,
                                 hipStream_t stream )
{
    hipLaunchKernelGGL( vrc_k_build_lut, dim3( 1 ), dim3( 256 ), 0, stream, tf, lut, p );
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * brick upload: row-major brick -> micro-blocked atlas (replaces the cudaMemcpy3DAsync into
 * a cudaArray of cuda/TexturePool.cu:187-201; the "array layout" is ours to define)
 * ---------------------------------------------------------------------------------------- */
/* fast path: 1-byte voxels, x extent a multiple of 8: one thread moves one 8-voxel run */
__global__ void vrc_k_repack_u8x8( const uint2* __restrict__ src, uint8_t* __restrict__ atlas,
                                   uint32_t sx8, uint32_t sy, uint32_t sz, uint32_t ox,
                                   uint32_t oy, uint32_t oz, uint32_t nbx, uint32_t nby )
{
    const uint32_t total = sx8 * sy * sz;
    for( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += gridDim.x * blockDim.x )
    {
        const uint32_t x8 = i % sx8;
        const uint32_t y = ( i / sx8 ) % sy;
        const uint32_t z = i / ( sx8 * sy );
        const uint2 v = src[i];
        const uint32_t e = vrc_swizzle( ox + x8 * 8u, oy + y, oz + z, nbx, nby );
        *reinterpret_cast< uint2* >( atlas + e ) = v;
    }
}

template < typename T >
__global__ void vrc_k_repack_generic( const T* __restrict__ src, T* __restrict__ atlas,
                                      uint32_t sx, uint32_t sy, uint32_t sz, uint32_t ox,
                                      uint32_t oy, uint32_t oz, uint32_t nbx, uint32_t nby )
{
    const size_t total = (size_t)sx * sy * sz;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % sx );
        const uint32_t y = (uint32_t)( ( i / sx ) % sy );
        const uint32_t z = (uint32_t)( i / ( (size_t)sx * sy ) );
        atlas[vrc_swizzle( ox + x, oy + y, oz + z, nbx, nby )] = src[i];
    }
}

template < typename T >
__global__ void vrc_k_read_region( const T* __restrict__ atlas, T* __restrict__ dst,
                                   uint32_t sx, uint32_t sy, uint32_t sz, uint32_t ox,
                                   uint32_t oy, uint32_t oz, uint32_t nbx, uint32_t nby )
{
    const size_t total = (size_t)sx * sy * sz;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % sx );
        const uint32_t y = (uint32_t)( ( i / sx ) % sy );
        const uint32_t z = (uint32_t)( i / ( (size_t)sx * sy ) );
        dst[i] = atlas[vrc_swizzle( ox + x, oy + y, oz + z, nbx, nby )];
    }
}

static uint32_t grid_for( size_t total, uint32_t block )
{
    size_t g = ( total + block - 1 ) / block;
    if( g > 2048 * 4 )
        g = 2048 * 4; /* grid-stride the rest (cdna_hip_programming.md guideline 11) */
    if( g == 0 )
        g = 1;
    return (uint32_t)g;
}

hipError_t vrc_launch_repack_brick( const void* src, void* atlas, uint32_t elemBytes,
                                    const uint32_t size[3], const uint32_t o[3], uint32_t nbx,
                                    uint32_t nby, hipStream_t stream )
{
    const size_t total = (size_t)size[0] * size[1] * size[2];
    if( total == 0 )
        return hipSuccess;
    if( elemBytes == 1 && ( size[0] % 8u ) == 0 && ( ( (uintptr_t)src ) % 8u ) == 0 &&
        ( o[0] % 8u ) == 0 && total / 8 < 0xFFFFFFFFull )
    {
        const uint32_t sx8 = size[0] / 8u;
        hipLaunchKernelGGL( vrc_k_repack_u8x8, dim3( grid_for( total / 8, 256 ) ), dim3( 256 ), 0,
                            stream, (const uint2*)src, (uint8_t*)atlas, sx8, size[1], size[2],
                            o[0], o[1], o[2], nbx, nby );
    }
    else if( elemBytes == 1 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint8_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint8_t*)src, (uint8_t*)atlas, size[0],
                            size[1], size[2], o[0], o[1], o[2], nbx, nby );
    else if( elemBytes == 2 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint16_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint16_t*)src, (uint16_t*)atlas,
                            size[0], size[1], size[2], o[0], o[1], o[2], nbx, nby );
    else if( elemBytes == 4 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint32_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint32_t*)src, (uint32_t*)atlas,
                            size[0], size[1], size[2], o[0], o[1], o[2], nbx, nby );
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t vrc_launch_read_region( const void* atlas, void* dst, uint32_t elemBytes,
                                   const uint32_t o[3], const uint32_t size[3], uint32_t nbx,
                                   uint32_t nby, hipStream_t stream )
{
    const size_t total = (size_t)size[0] * size[1] * size[2];
    if( total == 0 )
        return hipSuccess;
    const dim3 g( grid_for( total, 256 ) ), b( 256 );
    if( elemBytes == 1 )
        hipLaunchKernelGGL( vrc_k_read_region< uint8_t >, g, b, 0, stream, (const uint8_t*)atlas,
                            (uint8_t*)dst, size[0], size[1], size[2], o[0], o[1], o[2], nbx, nby );
    else if( elemBytes == 2 )
        hipLaunchKernelGGL( vrc_k_read_region< uint16_t >, g, b, 0, stream,
                            (const uint16_t*)atlas, (uint16_t*)dst, size[0], size[1], size[2],
                            o[0], o[1], o[2], nbx, nby );
    else if( elemBytes == 4 )
        hipLaunchKernelGGL( vrc_k_read_region< uint32_t >, g, b, 0, stream,
                            (const uint32_t*)atlas, (uint32_t*)dst, size[0], size[1], size[2],
                            o[0], o[1], o[2], nbx, nby );
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * the raycast kernel
 * ---------------------------------------------------------------------------------------- */
template < bool DDA, bool CLAMP, bool COUNT >
__global__ __launch_bounds__( VRC_WG ) void vrc_k_raycast(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes,
    const int32_t* __restrict__ gridTable, const uint8_t* __restrict__ atlas,
    const vrc_f4* __restrict__ lutGlobal, vrc_f4* __restrict__ pixelBuffer,
    unsigned long long* __restrict__ sampleCounter, const uint32_t tilesX, const uint32_t nTiles )
{
    __shared__ vrc_f4 lut[256];
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for( uint32_t i = 0; i < 256u / VRC_WG; ++i )
        lut[lane + i * VRC_WG] = lutGlobal[lane + i * VRC_WG];
    __syncthreads();

    /* XCD-aware remap: workgroups b and b+8 run on the same XCD (MI355X_MICROARCH.md,
     * "Workgroup dispatch"); give XCD k the k-th contiguous run of tiles (row-major), so
     * each XCD's L2 serves one horizontal band of the image.  Bijective for any nTiles. */
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, j = b >> 3;
    const uint32_t per = nTiles >> 3, rem = nTiles & 7u;
    const uint32_t tile = xcd * per + ( xcd < rem ? xcd : rem ) + j;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t px = tx * VRC_TILE + ( lane & 7u );
    const uint32_t py = ty * VRC_TILE + ( lane >> 3 );

    uint32_t nSamples = 0;
    if( px < f.width && py < f.height )
    {
        if( DDA )
            vrc_pixel_grid_dda< CLAMP, COUNT, uint8_t >( f, nodes, gridTable, atlas, lut,
                                                         pixelBuffer, px, py, nSamples );
        else
            vrc_pixel_reference_order< CLAMP, COUNT, uint8_t >( f, nodes, atlas, lut, pixelBuffer,
                                                                px, py, nSamples );
    }
    if( COUNT )
    {
        /* wave64 reduction, one atomic per wave (guideline 12) */
        unsigned long long s = nSamples;
#pragma unroll
        for( int off = 32; off > 0; off >>= 1 )
            s += __shfl_down( s, off, 64 );
        if( lane == 0 && s != 0 )
            atomicAdd( sampleCounter, s );
    }
}

template < bool DDA, bool CLAMP, bool COUNT >
static hipError_t launch_variant( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + VRC_TILE - 1 ) / VRC_TILE;
    const uint32_t tilesY = ( a.frame.height + VRC_TILE - 1 ) / VRC_TILE;
    const uint32_t nTiles = tilesX * tilesY;
    if( nTiles == 0 )
        return hipSuccess;
    hipLaunchKernelGGL( ( vrc_k_raycast< DDA, CLAMP, COUNT > ), dim3( nTiles ), dim3( VRC_WG ), 0,
                        stream, a.frame, a.nodes, a.gridTable, (const uint8_t*)a.atlas, a.lut,
                        a.pixelBuffer, a.sampleCounter, tilesX, nTiles );
    return hipGetLastError();
}

hipError_t vrc_launch_raycast( const vrc_raycast_args& a, hipStream_t stream )
{
    const bool count = a.sampleCounter != nullptr;
    const int key = ( a.gridDda ? 4 : 0 ) | ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 );
    switch( key )
    {
    case 0: return launch_variant< false, false, false >( a, stream );
    case 1: return launch_variant< false, false, true >( a, stream );
    case 2: return launch_variant< false, true, false >( a, stream );
    case 3: return launch_variant< false, true, true >( a, stream );
    case 4: return launch_variant< true, false, false >( a, stream );
    case 5: return launch_variant< true, false, true >( a, stream );
    case 6: return launch_variant< true, true, false >( a, stream );
    default: return launch_variant< true, true, true >( a, stream );
    }
}
