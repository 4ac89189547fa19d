/*
 * vrc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the volume raycaster.
 *
 * The per-ray hot path of renderers/cudaRaycaster/cuda/Renderer.cu:95-230 re-designed for
 * CDNA4:
 *   - one wave64 = one 8x8 pixel tile, one workgroup = one wave, so the hardware
 *     dispatcher load-balances tiles (ray lengths differ by >2x across the image) and
 *     no barrier is needed after the table is staged;
 *   - workgroup -> tile mapping: units of 2x2 tiles heaviest-first, dealt to the XCDs in snake order (the
 *     dispatcher deals workgroups round-robin over the 8 XCDs; vrc_internal.h, "Tile schedule": spatial
 *     super-tiles per XCD were measured in round 4 -- 7 % fewer HBM requests, no time -- and one contiguous
 *     band per XCD in round 1, slower);
 *   - the atlas is read as 8x8x8-voxel micro-blocks (one z-slice of a block = one 64-byte
 *     segment), so the 64 fetches of a wave step land in a handful of cache lines;
 *   - TF lookup + opacity correction are folded into a 256-entry classified table staged
 *     in LDS once per workgroup (4 KiB): no per-sample pow, one ds_read_b128 per sample;
 *   - bricks are enumerated by a DDA over the brick grid instead of the O(nodes) loop;
 *   - no MFMA: this is byte gather + scalar compositing, there is no contraction.
 */
#include "vrc_internal.h"

#include <algorithm>

/* pixel tile of one wave: VRC_TILE_W x VRC_TILE_H = 64 (vrc_internal.h) */
#define VRC_WG 64u
/* waves (= tiles) per workgroup: they share the LDS tables, set up once per workgroup */
#ifndef VRC_WAVES_PER_WG
#define VRC_WAVES_PER_WG VRC_WAVES_PER_GROUP /* the four tiles of a schedule unit (vrc_internal.h) */
#endif
#define VRC_WG_THREADS ( VRC_WG * VRC_WAVES_PER_WG )

/* classified-sample table (vrc_core.h: vrc_lut_entry): 256 entries per frame instead of a TF
 * fetch + pow per sample */
__global__ void vrc_k_build_lut( const float* __restrict__ tf, vrc_f4* __restrict__ lut,
                                 vrc_lut_params p, int linear )
{
    const uint32_t d = threadIdx.x;
    if( linear )
    {
        /* trilinear filter: densities are continuous, the kernel classifies per sample from
         * the transfer function padded by one entry on both sides (vrc_classify) */
        for( uint32_t k = d; k < VRC_TFP_ENTRIES; k += blockDim.x )
        {
            const uint32_t i = k == 0u ? 0u : ( k - 1u > 255u ? 255u : k - 1u );
            const vrc_f4 e = { tf[i * 4u], tf[i * 4u + 1u], tf[i * 4u + 2u], tf[i * 4u + 3u] };
            lut[k] = e;
        }
        return;
    }
    if( d < 256u )
        lut[d] = vrc_lut_entry( tf, d, p );
    if( d == 0 )
    {
        /* entry 256: the no-op sample (vrc_march_segment) */
        const vrc_f4 z = { 0.f, 0.f, 0.f, 0.f };
        lut[256] = z;
        lut[257] = z;
    }
}

hipError_t vrc_launch_build_lut( const float* tf, vrc_f4* lut, vrc_lut_params p, bool linear,
                                 hipStream_t stream )
{
    hipLaunchKernelGGL( vrc_k_build_lut, dim3( 1 ), dim3( 256 ), 0, stream, tf, lut, p,
                        linear ? 1 : 0 );
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * brick upload: row-major brick -> micro-blocked atlas (replaces the cudaMemcpy3DAsync into
 * a cudaArray of cuda/TexturePool.cu:187-201; the "array layout" is ours to define)
 * ---------------------------------------------------------------------------------------- */
/* fast path: 1-byte voxels, x extent a multiple of 8: one thread moves one 8-voxel run
 * (= one row of a micro-block z-slice, 8-byte aligned on both sides) */
__global__ void vrc_k_repack_u8x8( const uint2* __restrict__ src, uint8_t* __restrict__ slot,
                                   uint32_t sx8, uint32_t sy, uint32_t sz, uint32_t sbx,
                                   uint32_t sby )
{
    const uint32_t total = sx8 * sy * sz;
    for( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += gridDim.x * blockDim.x )
    {
        const uint32_t x8 = i % sx8;
        const uint32_t y = ( i / sx8 ) % sy;
        const uint32_t z = i / ( sx8 * sy );
        const uint2 v = src[i];
        const uint32_t e = vrc_slot_local_index( x8 * 8u, y, z, sbx, sby );
        *reinterpret_cast< uint2* >( slot + e ) = v;
    }
}

template < typename T >
__global__ void vrc_k_repack_generic( const T* __restrict__ src, T* __restrict__ slot,
                                      uint32_t sx, uint32_t sy, uint32_t sz, uint32_t sbx,
                                      uint32_t sby )
{
    const size_t total = (size_t)sx * sy * sz;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % sx );
        const uint32_t y = (uint32_t)( ( i / sx ) % sy );
        const uint32_t z = (uint32_t)( i / ( (size_t)sx * sy ) );
        slot[vrc_slot_local_index( x, y, z, sbx, sby )] = src[i];
    }
}

/* brick smaller than its (8-voxel padded) slot: the slot padding gets the brick's border
 * voxels, i.e. clamp addressing at the brick border is baked into the data -- what the
 * clamped trilinear taps one voxel past the brick must read (cudaAddressModeClamp on a
 * texture that ends with the brick, cuda/TexturePool.cu:163-170) */
template < typename T >
__global__ void vrc_k_repack_padded( const T* __restrict__ src, T* __restrict__ slot, uint32_t sx,
                                     uint32_t sy, uint32_t sz, uint32_t dx, uint32_t dy,
                                     uint32_t dz )
{
    const size_t total = (size_t)dx * dy * dz;
    const uint32_t sbx = dx >> VRC_MB_SHIFT, sby = dy >> VRC_MB_SHIFT;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % dx );
        const uint32_t y = (uint32_t)( ( i / dx ) % dy );
        const uint32_t z = (uint32_t)( i / ( (size_t)dx * dy ) );
        const uint32_t cx = x < sx ? x : sx - 1u, cy = y < sy ? y : sy - 1u, cz = z < sz ? z : sz - 1u;
        slot[vrc_slot_local_index( x, y, z, sbx, sby )] = src[( (size_t)cz * sy + cy ) * sx + cx];
    }
}

template < typename T >
__global__ void vrc_k_read_region( const T* __restrict__ atlas, T* __restrict__ dst,
                                   uint32_t sx, uint32_t sy, uint32_t sz, uint32_t ox,
                                   uint32_t oy, uint32_t oz, const vrc_layout lay )
{
    const size_t total = (size_t)sx * sy * sz;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % sx );
        const uint32_t y = (uint32_t)( ( i / sx ) % sy );
        const uint32_t z = (uint32_t)( i / ( (size_t)sx * sy ) );
        dst[i] = atlas[vrc_atlas_index( lay, ox + x, oy + y, oz + z )];
    }
}

/* ------------------------------------------------------------------------------------------
 * brick histogram, a side kernel on a resident brick (livre/lib/cache/HistogramObject.cpp:36-119
 * computes it on the CPU from the data cache): interior voxels only (the overlap is skipped,
 * :94-97), integral types are binned over the type's range (:48-52, :104-110), every voxel
 * counts scaleFactor times (:111).  One LDS histogram per workgroup, then 64-bit global adds.
 * ---------------------------------------------------------------------------------------- */
template < typename T >
__global__ __launch_bounds__( 256 ) void vrc_k_brick_histogram(
    const T* __restrict__ slot, uint32_t sbx, uint32_t sby, uint32_t ox, uint32_t oy, uint32_t oz,
    uint32_t nx, uint32_t ny, uint32_t nz, uint32_t binCount, uint32_t perBin,
    unsigned long long scale, unsigned long long* __restrict__ bins )
{
    __shared__ uint32_t h[4096];
    for( uint32_t i = threadIdx.x; i < binCount; i += blockDim.x )
        h[i] = 0;
    __syncthreads();
    const size_t total = (size_t)nx * ny * nz;
    for( size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x )
    {
        const uint32_t x = (uint32_t)( i % nx ), y = (uint32_t)( ( i / nx ) % ny ),
                       z = (uint32_t)( i / ( (size_t)nx * ny ) );
        const uint32_t v = (uint32_t)slot[vrc_slot_local_index( ox + x, oy + y, oz + z, sbx, sby )];
        atomicAdd( &h[v / perBin], 1u );
    }
    __syncthreads();
    for( uint32_t i = threadIdx.x; i < binCount; i += blockDim.x )
        if( h[i] )
            atomicAdd( &bins[i], (unsigned long long)h[i] * scale );
}

hipError_t vrc_launch_brick_histogram( const void* slot, uint32_t elemBytes, uint32_t sbx, uint32_t sby,
                                       const uint32_t origin[3], const uint32_t size[3],
                                       uint32_t binCount, unsigned long long scale,
                                       unsigned long long* bins, hipStream_t stream )
{
    const size_t total = (size_t)size[0] * size[1] * size[2];
    if( total == 0 || binCount == 0 || binCount > 4096 )
        return hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)std::min< size_t >( ( total + 255 ) / 256, 1024 );
    if( elemBytes == 1 && 256u % binCount == 0 )
        hipLaunchKernelGGL( vrc_k_brick_histogram< uint8_t >, dim3( blocks ), dim3( 256 ), 0, stream,
                            (const uint8_t*)slot, sbx, sby, origin[0], origin[1], origin[2], size[0],
                            size[1], size[2], binCount, 256u / binCount, scale, bins );
    else if( elemBytes == 2 && 65536u % binCount == 0 )
        hipLaunchKernelGGL( vrc_k_brick_histogram< uint16_t >, dim3( blocks ), dim3( 256 ), 0, stream,
                            (const uint16_t*)slot, sbx, sby, origin[0], origin[1], origin[2], size[0],
                            size[1], size[2], binCount, 65536u / binCount, scale, bins );
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
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

hipError_t vrc_launch_repack_brick( const void* src, void* slot, uint32_t elemBytes,
                                    const uint32_t size[3], const uint32_t slotDim[3],
                                    hipStream_t stream )
{
    const size_t total = (size_t)size[0] * size[1] * size[2];
    if( total == 0 )
        return hipSuccess;
    const uint32_t sbx = slotDim[0] >> VRC_MB_SHIFT, sby = slotDim[1] >> VRC_MB_SHIFT;
    if( size[0] != slotDim[0] || size[1] != slotDim[1] || size[2] != slotDim[2] )
    {
        const size_t all = (size_t)slotDim[0] * slotDim[1] * slotDim[2];
        const dim3 g( grid_for( all, 256 ) ), b( 256 );
        if( elemBytes == 1 )
            hipLaunchKernelGGL( vrc_k_repack_padded< uint8_t >, g, b, 0, stream, (const uint8_t*)src,
                                (uint8_t*)slot, size[0], size[1], size[2], slotDim[0], slotDim[1], slotDim[2] );
        else if( elemBytes == 2 )
            hipLaunchKernelGGL( vrc_k_repack_padded< uint16_t >, g, b, 0, stream, (const uint16_t*)src,
                                (uint16_t*)slot, size[0], size[1], size[2], slotDim[0], slotDim[1], slotDim[2] );
        else if( elemBytes == 4 )
            hipLaunchKernelGGL( vrc_k_repack_padded< uint32_t >, g, b, 0, stream, (const uint32_t*)src,
                                (uint32_t*)slot, size[0], size[1], size[2], slotDim[0], slotDim[1], slotDim[2] );
        else
            return hipErrorInvalidValue;
        return hipGetLastError();
    }
    /* (the experimental layout 5 has no 8-voxel runs along x: generic kernel) */
    if( VRC_LAYOUT != 5 && elemBytes == 1 && ( size[0] % 8u ) == 0 && ( ( (uintptr_t)src ) % 8u ) == 0 &&
        ( ( (uintptr_t)slot ) % 8u ) == 0 && total / 8 < 0xFFFFFFFFull )
    {
        const uint32_t sx8 = size[0] / 8u;
        hipLaunchKernelGGL( vrc_k_repack_u8x8, dim3( grid_for( total / 8, 256 ) ), dim3( 256 ), 0,
                            stream, (const uint2*)src, (uint8_t*)slot, sx8, size[1], size[2], sbx,
                            sby );
    }
    else if( elemBytes == 1 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint8_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint8_t*)src, (uint8_t*)slot, size[0],
                            size[1], size[2], sbx, sby );
    else if( elemBytes == 2 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint16_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint16_t*)src, (uint16_t*)slot, size[0],
                            size[1], size[2], sbx, sby );
    else if( elemBytes == 4 )
        hipLaunchKernelGGL( vrc_k_repack_generic< uint32_t >, dim3( grid_for( total, 256 ) ),
                            dim3( 256 ), 0, stream, (const uint32_t*)src, (uint32_t*)slot, size[0],
                            size[1], size[2], sbx, sby );
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t vrc_launch_read_region( const void* atlas, void* dst, uint32_t elemBytes,
                                   const uint32_t o[3], const uint32_t size[3],
                                   const vrc_layout& lay, hipStream_t stream )
{
    const size_t total = (size_t)size[0] * size[1] * size[2];
    if( total == 0 )
        return hipSuccess;
    const dim3 g( grid_for( total, 256 ) ), b( 256 );
    if( elemBytes == 1 )
        hipLaunchKernelGGL( vrc_k_read_region< uint8_t >, g, b, 0, stream, (const uint8_t*)atlas,
                            (uint8_t*)dst, size[0], size[1], size[2], o[0], o[1], o[2], lay );
    else if( elemBytes == 2 )
        hipLaunchKernelGGL( vrc_k_read_region< uint16_t >, g, b, 0, stream,
                            (const uint16_t*)atlas, (uint16_t*)dst, size[0], size[1], size[2],
                            o[0], o[1], o[2], lay );
    else if( elemBytes == 4 )
        hipLaunchKernelGGL( vrc_k_read_region< uint32_t >, g, b, 0, stream,
                            (const uint32_t*)atlas, (uint32_t*)dst, size[0], size[1], size[2],
                            o[0], o[1], o[2], lay );
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}


/* ------------------------------------------------------------------------------------------
 * tap-packed atlas (vrc_core.h, "tap-packed form of the trilinear filter"): texel (x,y,z) of a slot =
 * v[x,y,z] | v[x,y,z+1] << 8 (<< 16 for 16-bit voxels) of the atlas's slot, in blocks of 64 rows of 9 texels (vrc_packed_decode; the ninth column
 * repeats the next block's first; the z neighbour and the copy clamped at the slot's last voxel: never read with a
 * weight, a sample's lower tap is at most slotDim - 2 with overlap >= 1).  One thread per packed texel, coalesced
 * 2-byte stores; the byte reads hit L1/L2.
 * ---------------------------------------------------------------------------------------- */
template < typename V, typename T >
__global__ __launch_bounds__( 256 ) void vrc_k_pack_slots( const V* __restrict__ atlas, T* __restrict__ packed,
                                                           uint64_t firstSlot, uint32_t slotBlocks, uint32_t sdx, uint32_t sdy,
                                                           uint32_t sdz, uint32_t sbx, uint32_t sby )
{
    /* blockIdx.y = the slot (64-bit only in its base), x strides over the slot's packed texels in 32 bits */
    const uint64_t slotIndex = firstSlot + blockIdx.y;
    const V* const slot = atlas + slotIndex * ( (uint64_t)slotBlocks * VRC_MB_VOXELS );
    T* const out = packed + slotIndex * ( (uint64_t)slotBlocks * VRC_PK_BLOCK );
    const uint32_t n = slotBlocks * VRC_PK_BLOCK;
    (void)sdy;
    for( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x )
    {
        const uint32_t b = i / VRC_PK_BLOCK, in = i % VRC_PK_BLOCK;
        uint32_t ix, iy, iz;
        vrc_packed_decode( in, ix, iy, iz );
        uint32_t x = ( b % sbx ) * 8u + ix;
        const uint32_t y = ( ( b / sbx ) % sby ) * 8u + iy, z = ( b / ( sbx * sby ) ) * 8u + iz;
        x = x < sdx ? x : sdx - 1u;
        const uint32_t z1 = z + 1u < sdz ? z + 1u : z;
        out[i] = vrc_pack_taps< T >( slot[vrc_slot_local_index( x, y, z, sbx, sby )], slot[vrc_slot_local_index( x, y, z1, sbx, sby )] );
    }
}

hipError_t vrc_launch_pack_slots( const void* atlas, void* packed, uint64_t firstElem, uint64_t nElems,
                                  const uint32_t slotDim[3], uint32_t elemBytes, hipStream_t stream )
{
    if( nElems == 0 )
        return hipSuccess;
    if( elemBytes != 1u && elemBytes != 2u )
        return hipErrorInvalidValue;
    const uint32_t slotBlocks = ( slotDim[0] >> VRC_MB_SHIFT ) * ( slotDim[1] >> VRC_MB_SHIFT ) * ( slotDim[2] >> VRC_MB_SHIFT );
    const uint64_t slotElems = (uint64_t)slotBlocks * VRC_MB_VOXELS;
    const uint64_t firstSlot = firstElem / slotElems, nSlots = nElems / slotElems;
    const uint32_t gx = grid_for( (size_t)slotBlocks * VRC_PK_BLOCK, 256 );
    /* (grid.y holds at most 65535 slots per launch) */
    for( uint64_t s0 = 0; s0 < nSlots; s0 += 65535u )
    {
        const uint32_t ns = (uint32_t)std::min< uint64_t >( nSlots - s0, 65535u );
        const dim3 grid( nSlots > 64u ? std::min( gx, 64u ) : gx, ns );
        if( elemBytes == 1u )
            hipLaunchKernelGGL( ( vrc_k_pack_slots< uint8_t, uint16_t > ), grid, dim3( 256 ), 0, stream, (const uint8_t*)atlas,
                                (uint16_t*)packed, firstSlot + s0, slotBlocks, slotDim[0], slotDim[1], slotDim[2],
                                slotDim[0] >> VRC_MB_SHIFT, slotDim[1] >> VRC_MB_SHIFT );
        else
            hipLaunchKernelGGL( ( vrc_k_pack_slots< uint16_t, uint32_t > ), grid, dim3( 256 ), 0, stream, (const uint16_t*)atlas,
                                (uint32_t*)packed, firstSlot + s0, slotBlocks, slotDim[0], slotDim[1], slotDim[2],
                                slotDim[0] >> VRC_MB_SHIFT, slotDim[1] >> VRC_MB_SHIFT );
    }
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * tile schedule (vrc_internal.h): super-tiles of VRC_SUPER_UNITS^2 units ordered by estimated work, heaviest first
 * -- cost = chord of the super-tile-centre ray through the (clipped) volume box, counting sort over 256 cost buckets
 * -- and dealt to the XCDs in snake order; workgroup b = unit b / 8 of XCD b % 8's list.  Re-run only when the view
 * changes.
 * ---------------------------------------------------------------------------------------- */
__device__ __forceinline__ uint32_t vrc_tile_bucket( const vrc_frame& f, uint32_t super, uint32_t superX, float invDiag )
{
    /* cost at the super-tile's centre (clamped into the frame) */
    const uint32_t span = VRC_SUPER_UNITS * 2u;
    const uint32_t sx = super % superX, sy = super / superX;
    uint32_t px = sx * VRC_TILE_W * span + VRC_TILE_W * span / 2u, py = sy * VRC_TILE_H * span + VRC_TILE_H * span / 2u;
    px = px < f.width ? px : f.width - 1;
    py = py < f.height ? py : f.height - 1;
    const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
    if( !r.hit )
        return 255u;
    const float chord = r.tFarGlobal - fmaxf( r.tNearGlobal, r.tNearPlane );
    const float q = fminf( fmaxf( chord * invDiag, 0.0f ), 1.0f );
    return 255u - (uint32_t)( q * 255.0f );
}

/* pass 1: cost bucket of every super-tile, histogram; the last workgroup to finish turns the
 * histogram into bucket start offsets.  scratch: [0..255] histogram/offsets, [256] counter
 * of finished workgroups (zeroed by the launcher). */
__global__ __launch_bounds__( 256 ) void vrc_k_tile_bucket( const vrc_frame f, const uint32_t superX,
                                                            const uint32_t nSuper,
                                                            uint8_t* __restrict__ bucket,
                                                            uint32_t* __restrict__ scratch )
{
    __shared__ uint32_t hist[256];
    __shared__ bool last;
    const float dx = f.aabbMax[0] - f.aabbMin[0], dy = f.aabbMax[1] - f.aabbMin[1],
                dz = f.aabbMax[2] - f.aabbMin[2];
    const float invDiag = 1.0f / sqrtf( dx * dx + dy * dy + dz * dz );
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if( t < nSuper )
    {
        const uint32_t b = vrc_tile_bucket( f, t, superX, invDiag );
        bucket[t] = (uint8_t)b;
        atomicAdd( &hist[b], 1u );
    }
    __syncthreads();
    if( hist[threadIdx.x] )
        atomicAdd( &scratch[threadIdx.x], hist[threadIdx.x] );
    __threadfence();
    __syncthreads();
    if( threadIdx.x == 0 )
        last = atomicAdd( &scratch[256], 1u ) == gridDim.x - 1u;
    __syncthreads();
    if( !last )
        return;
    /* exclusive scan of the 256 bucket counts (all workgroups' adds are visible: fence above) */
    hist[threadIdx.x] = __hip_atomic_load( &scratch[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    __syncthreads();
    for( uint32_t off = 1; off < 256u; off <<= 1 )
    {
        const uint32_t v = threadIdx.x >= off ? hist[threadIdx.x - off] : 0u;
        __syncthreads();
        hist[threadIdx.x] += v;
        __syncthreads();
    }
    scratch[threadIdx.x] = threadIdx.x ? hist[threadIdx.x - 1u] : 0u;
}

/* pass 2: every super-tile takes its rank (bucket start + a rank inside the bucket: order inside a bucket is
 * arbitrary), the rank its XCD and its place in that XCD's list, and writes the tiles of its units there.  A
 * workgroup reserves its share of every bucket with one global atomic per bucket; the ranks inside the share come
 * from LDS atomics.  `order` was filled with VRC_NO_TILE by the launcher (places no super-tile takes). */
__global__ __launch_bounds__( 256 ) void vrc_k_tile_scatter( const uint32_t nSuper, const uint32_t superX,
                                                             const uint8_t* __restrict__ bucket,
                                                             uint32_t* __restrict__ scratch,
                                                             uint32_t* __restrict__ order,
                                                             const uint32_t frameTilesX, const uint32_t frameTilesY )
{
    __shared__ uint32_t cnt[256], base[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t b = 0, rank = 0;
    if( t < nSuper )
    {
        b = bucket[t];
        rank = atomicAdd( &cnt[b], 1u );
    }
    __syncthreads();
    if( cnt[threadIdx.x] )
        base[threadIdx.x] = atomicAdd( &scratch[threadIdx.x], cnt[threadIdx.x] );
    __syncthreads();
    if( t < nSuper )
    {
        const uint32_t r = base[b] + rank;
        /* snake over the XCDs: ranks 0..7 go to XCDs 0..7, ranks 8..15 to XCDs 7..0, ... */
        const uint32_t row = r / VRC_XCDS, col = r % VRC_XCDS;
        const uint32_t xcd = ( row & 1u ) ? VRC_XCDS - 1u - col : col;
        const uint32_t unitsX = ( frameTilesX + 1u ) / 2u, unitsY = ( frameTilesY + 1u ) / 2u;
        const uint32_t sx = t % superX, sy = t / superX;
        for( uint32_t u = 0; u < VRC_SUPER_UNITS * VRC_SUPER_UNITS; ++u )
        {
            /* Morton order inside the super-tile (up to 16 x 16 units) */
            const uint32_t ux = ( u & 1u ) | ( ( u >> 1 ) & 2u ) | ( ( u >> 2 ) & 4u ) | ( ( u >> 3 ) & 8u );
            const uint32_t uy = ( ( u >> 1 ) & 1u ) | ( ( u >> 2 ) & 2u ) | ( ( u >> 3 ) & 4u ) | ( ( u >> 4 ) & 8u );
            const uint32_t x = sx * VRC_SUPER_UNITS + ux, y = sy * VRC_SUPER_UNITS + uy;
            /* the workgroup: VRC_WAVES_PER_WG / 4 consecutive units of the super-tile's Morton order (one in the product) */
            /* (developer builds with fewer than four waves per workgroup: the units in rank order, as in rounds 1-3) */
            constexpr uint32_t U = VRC_WAVES_PER_WG >= 4u ? VRC_WAVES_PER_WG / 4u : 1u;
            const uint32_t g = VRC_WAVES_PER_WG >= 4u ? VRC_XCDS * ( row * ( VRC_SUPER_UNITS * VRC_SUPER_UNITS / U ) + u / U ) + xcd
                                                      : r * VRC_SUPER_UNITS * VRC_SUPER_UNITS + u;
#pragma unroll
            for( uint32_t sub = 0; sub < 4u; ++sub )
                order[( g * U + u % U ) * 4u + sub] = ( x < unitsX && y < unitsY ) ? vrc_unit_tile( y * unitsX + x, sub, frameTilesX, frameTilesY )
                                                                                 : VRC_NO_TILE;
        }
    }
}

hipError_t vrc_launch_tile_order( const vrc_frame& f, uint32_t* order, uint32_t* scratch,
                                  uint8_t* bucket, hipStream_t stream )
{
    const uint32_t frameTilesX = ( f.width + VRC_TILE_W - 1 ) / VRC_TILE_W;
    const uint32_t frameTilesY = ( f.height + VRC_TILE_H - 1 ) / VRC_TILE_H;
    const uint32_t superX = vrc_super_x( frameTilesX ), nSuper = superX * vrc_super_x( frameTilesY );
    if( frameTilesX * frameTilesY == 0 )
        return hipSuccess;
    hipError_t e = hipMemsetAsync( scratch, 0, VRC_TILE_SCRATCH_WORDS * sizeof( uint32_t ), stream );
    if( e == hipSuccess )
        e = hipMemsetAsync( order, 0xFF, (size_t)vrc_schedule_slots( frameTilesX, frameTilesY ) * sizeof( uint32_t ), stream );
    if( e != hipSuccess )
        return e;
    const dim3 grid( ( nSuper + 255u ) / 256u ), block( 256 );
    hipLaunchKernelGGL( vrc_k_tile_bucket, grid, block, 0, stream, f, superX, nSuper, bucket, scratch );
    hipLaunchKernelGGL( vrc_k_tile_scatter, grid, block, 0, stream, nSuper, superX, bucket, scratch, order, frameTilesX,
                        frameTilesY );
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * the raycast kernel
 * ---------------------------------------------------------------------------------------- */
#ifndef VRC_GREY_GROUP
/* samples a lane of the grey form keeps in flight: its two-float colours and table entries leave registers for 14 at
 * five waves per SIMD (93 VGPRs with the sample counter, 82 without); measured on C2 against 8: 0.509 -> 0.481 ms
 * (mem://), 0.512 -> 0.482 ms (noise); 12: 0.491 / 0.484; 16 (96 VGPRs): 0.487 */
#define VRC_GREY_GROUP 14
#endif
#ifndef VRC_PACKED_WAVES
#define VRC_PACKED_WAVES 3 /* (developer switch; 2 ... 6 measured on C2 with groups of 4 ... 24: the group decides, not the waves) */
#endif
#if defined( VRC_WG_TIMELINE )
/* developer build (tools/dev_timeline.py): when every wave of the last launch started and ended
 * (s_memrealtime: one 100 MHz clock for the whole chip) and where it ran (HW_ID, XCC_ID) */
struct vrc_wave_stamp
{
    unsigned long long start, end;
    uint32_t hwId, xccId;
};
__device__ vrc_wave_stamp vrc_timeline[1u << 16];
extern "C" int vrc_dev_read_timeline( void* out, size_t bytes )
{
    return (int)hipMemcpyFromSymbol( out, HIP_SYMBOL( vrc_timeline ), bytes < sizeof( vrc_timeline ) ? bytes : sizeof( vrc_timeline ) );
}
#endif
#ifndef VRC_MIN_WAVES
#define VRC_MIN_WAVES 5 /* measured on C2: 4 -> 5 waves per SIMD with four-wave workgroups: -2 % */
#endif
/* GROUP: samples a lane keeps in flight.  8 is fastest when the launch fills the machine several
 * times over (4 waves per SIMD); 16 (2 waves per SIMD) halves the dependent round trips of a ray
 * and wins when a launch is no more than about one wave per SIMD slot and the longest ray's
 * latency sets the time -- the per-rank share of a sort-first frame from 4 ranks up. */
template < bool DDA, bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP,
           bool BIG = false >
/* waves per SIMD the compiler plans for (at least): 5 for the table-driven point-sampling instances; the per-sample
 * classification modes, the float position chain, the clamped sampler and 64-bit slot bases need more registers than
 * 5 waves leave (they spilled at 5: the trilinear gather form ran 5.3 instead of 2.9 ms).
 * And at MOST: the grey form in groups of 14 needs 80 registers, a sixth wave per SIMD would fit, and six thrash the
 * L1 (frames in flight: 2180 -> 1730 frames/s).  amdgpu_waves_per_eu( min, max ) says so to the
 * compiler, which then reports the register count that admits no sixth wave (88); rounds 2-3 said it with 20 KiB of
 * LDS the kernel never touched. */
#define VRC_K_MIN_WAVES                                                                                             \
    ( ( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )                                                   \
          ? VRC_PACKED_WAVES                                                                                        \
          : GROUP > VRC_GREY_GROUP                                                                                  \
                ? 2                                                                                                 \
                : ( ( ( MODE == VRC_MODE_TABLE && GROUP <= 8 ) || MODE == VRC_MODE_GREY ) && FIXED && !CLAMP && !BIG \
                        ? VRC_MIN_WAVES                                                                             \
                        : ( GROUP > 8 ? 2 : 4 ) ) )
#define VRC_K_MAX_WAVES ( ( MODE == VRC_MODE_GREY && GROUP <= VRC_GREY_GROUP && VRC_WAVES_PER_WG == 4u ) ? VRC_GREY_MAX_WAVES : 8 )
#ifndef VRC_GREY_MAX_WAVES
#define VRC_GREY_MAX_WAVES 5
#endif
__global__ __launch_bounds__( VRC_WG_THREADS ) __attribute__( ( amdgpu_waves_per_eu( VRC_K_MIN_WAVES, VRC_K_MAX_WAVES ) ) ) void vrc_k_raycast(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes,
    const int32_t* __restrict__ gridTable, const ATLAS_T* __restrict__ atlas,
    const vrc_f4* __restrict__ lutGlobal, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles )
{
    /* classified table (257 entries) or, for the per-sample classification modes, the padded
     * transfer function (258) */
    __shared__ vrc_f4 lut[VRC_CLS8_ENTRIES];
    __shared__ uint16_t vrc_tile_cand[DDA ? 1u : VRC_WAVES_PER_WG * VRC_TILE_CANDIDATES];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
#if defined( VRC_WG_TIMELINE )
    struct vrc_stamp_scope
    {
        uint32_t index;
        bool on;
        unsigned long long t0;
        __device__ vrc_stamp_scope( uint32_t i, bool o ) : index( i & 0xFFFFu ), on( o ), t0( wall_clock64() ) {}
        __device__ ~vrc_stamp_scope()
        {
            if( on )
            {
                vrc_timeline[index].start = t0;
                vrc_timeline[index].end = wall_clock64();
                vrc_timeline[index].hwId = __builtin_amdgcn_s_getreg( ( 31 << 11 ) | 4 );
                vrc_timeline[index].xccId = __builtin_amdgcn_s_getreg( ( 3 << 11 ) | 20 );
            }
        }
    } vrc_stamp( blockIdx.x * VRC_WAVES_PER_WG + ( tid >> 6 ), lane == 0u );
#endif
    if( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
    {
        /* the packed march's classifier table (vrc_cls8_entry) from the padded transfer function */
        for( uint32_t i = tid; i < VRC_CLS8_ENTRIES; i += VRC_WG_THREADS )
            lut[i] = vrc_cls8_entry( lutGlobal, i, MODE == VRC_MODE_PACKED_GREY );
    }
    else if( MODE == VRC_MODE_GREY || MODE == VRC_MODE_POINT_GREY )
    {
        /* grey table: (rgb * alpha', alpha') as two floats per entry, 257 entries -- or, for the per-sample
         * classification of 16-bit voxels, the 258 entries of the padded transfer function as (grey, alpha) */
        vrc_f2* const lut2 = reinterpret_cast< vrc_f2* >( lut );
        for( uint32_t i = tid; i < ( MODE == VRC_MODE_GREY ? VRC_LUT_ENTRIES : VRC_TFP_ENTRIES ); i += VRC_WG_THREADS )
        {
            const vrc_f4 e = lutGlobal[i];
            lut2[i] = vrc_f2{ e.x, e.w };
        }
    }
    else
    {
#pragma unroll
        for( uint32_t i = 0; i < ( 256u + VRC_WG_THREADS - 1u ) / VRC_WG_THREADS; ++i )
            if( tid + i * VRC_WG_THREADS < 256u )
                lut[tid + i * VRC_WG_THREADS] = lutGlobal[tid + i * VRC_WG_THREADS];
        if( tid < VRC_TFP_ENTRIES - 256u )
            lut[256u + tid] = lutGlobal[256u + tid];
    }
#if defined( VRC_ADDR_TABLES )
    if( MODE == VRC_MODE_PACKED || MODE == VRC_MODE_PACKED_GREY )
    {
        /* per-axis BYTE offsets of the packed atlas's texels (vrc_core.h: vrc_pk_x / y / z) */
        for( uint32_t u = tid; u < 256u; u += VRC_WG_THREADS )
        {
            constexpr uint32_t TB = sizeof( ATLAS_T ) == 8 ? 4u : 2u; /* bytes per texel (ATLAS_T: the packed mode's tag) */
            vrc_addr_tab[u] = TB * vrc_pk_x( u );
            vrc_addr_tab[256u + u] = TB * vrc_pk_y( u, f.sbx );
            vrc_addr_tab[512u + u] = TB * vrc_pk_z( u, f.sbx, f.sby );
        }
    }
    else if( FIXED )
    {
        const vrc_lay lay = vrc_make_lay( f.sbx, f.sby );
#pragma unroll
        for( uint32_t i = 0; i < ( 256u + VRC_WG_THREADS - 1u ) / VRC_WG_THREADS; ++i )
        {
            const uint32_t u = tid + i * VRC_WG_THREADS;
            if( u < 256u )
            {
                vrc_addr_tab[u] = vrc_lay_x( lay, u );
                vrc_addr_tab[256u + u] = vrc_lay_y( lay, u );
                vrc_addr_tab[512u + u] = vrc_lay_z( lay, u );
            }
        }
    }
#endif
    __syncthreads();
    /* from here on the waves of the workgroup are independent */
    const uint32_t slotIndex = blockIdx.x * VRC_WAVES_PER_WG + ( tid >> 6 );
    const uint32_t tilesY = nTiles / tilesX;
    if( slotIndex >= vrc_schedule_slots( tilesX, tilesY ) )
        return;

    /* Workgroup -> tile: the schedule of vrc_internal.h.  Ray lengths vary by more than 2x over the image and
     * whole tiles miss the volume, so the heaviest units start first, dealt evenly to the XCDs (the dispatcher
     * deals workgroups b, b+1, ... round-robin over the 8 XCDs, MI355X_MICROARCH.md "Workgroup dispatch"). */
    const uint32_t tile = vrc_slot_tile( tileOrder, slotIndex, tilesX, tilesY );
    if( tile == VRC_NO_TILE )
        return;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
#if defined( VRC_LANES_ROWMAJOR ) || VRC_TILE_W != 8
    const uint32_t lx = lane % VRC_TILE_W, ly = lane / VRC_TILE_W;
#else
    /* Morton lane order: every 4 consecutive lanes (the unit the texture addresser works on)
     * are a 2x2 pixel quad, so their voxels usually share one 64-byte segment */
    const uint32_t lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
    const uint32_t ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
#endif
    const uint32_t px = tx * VRC_TILE_W + lx;
    const uint32_t py = ty * VRC_TILE_H + ly;

    uint32_t nSamples = 0;
    /* reference-order loop: the bricks this tile's rays can hit at all, found once per wave (vrc_core.h, "tile
     * culling"): 64 bricks per step against the pyramid through the tile's corners */
    const uint16_t* cand = nullptr;
    uint32_t nCand = 0;
    if( !DDA && f.nodeCount > 8u && f.nodeCount <= 65535u )
    {
        uint16_t* const mine = vrc_tile_cand + ( tid >> 6 ) * VRC_TILE_CANDIDATES;
        /* frame rows of the tile (row bands: local rows map to frame rows, not always consecutive ones) */
        float r0 = 1e30f, r1 = -1e30f;
        for( uint32_t j = 0; j < VRC_TILE_H; ++j )
        {
            const uint32_t ly = ty * VRC_TILE_H + j;
            if( ly < f.height )
            {
                const float fr = (float)( f.rowMap ? f.rowMap[ly] : ly );
                r0 = fminf( r0, fr );
                r1 = fmaxf( r1, fr );
            }
        }
        const vrc_tile_pyramid pyr = vrc_make_tile_pyramid( f, (float)( tx * VRC_TILE_W ) - 1.0f, r0 - 1.0f,
                                                           (float)( tx * VRC_TILE_W + VRC_TILE_W ), r1 + 1.0f );
        for( uint32_t base = 0; base < f.nodeCount; base += 64u )
        {
            const uint32_t i = base + lane;
            bool in = false;
            if( i < f.nodeCount )
                in = vrc_pyramid_may_hit( pyr, nodes[i].aabbMin, nodes[i].aabbSize );
            const uint64_t m = __builtin_amdgcn_ballot_w64( in );
            const uint32_t at = nCand + __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ),
                                                                   __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0u ) );
            if( in && at < VRC_TILE_CANDIDATES )
                mine[at] = (uint16_t)i;
            nCand += (uint32_t)__builtin_popcountll( m );
        }
        __builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" );
        __builtin_amdgcn_wave_barrier();
        if( nCand <= VRC_TILE_CANDIDATES )
            cand = mine;
    }
    if( px < f.width && py < f.height )
    {
        if( DDA )
            vrc_pixel_grid_dda< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >(
                f, nodes, gridTable, atlas, lut, cls, pixelBuffer, px, py, nSamples );
        else
            vrc_pixel_reference_order< CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >(
                f, nodes, atlas, lut, cls, pixelBuffer, px, py, nSamples, cand, nCand );
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

template < bool DDA, bool CLAMP, bool COUNT, bool FIXED, int MODE, typename ATLAS_T, int GROUP = VRC_GROUP,
           bool BIG = false >
static hipError_t launch_variant( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + VRC_TILE_W - 1 ) / VRC_TILE_W;
    const uint32_t tilesY = ( a.frame.height + VRC_TILE_H - 1 ) / VRC_TILE_H;
    const uint32_t nTiles = tilesX * tilesY;
    if( nTiles == 0 )
        return hipSuccess;
    vrc_internal_note_kernel( "vrc_k_raycast<%s,%s,%s,%s,%d,%s,%d,%s>", DDA ? "true" : "false", CLAMP ? "true" : "false",
                              COUNT ? "true" : "false", FIXED ? "true" : "false", (int)MODE,
                              sizeof( ATLAS_T ) == 1 ? "unsigned char" : ( sizeof( ATLAS_T ) == 2 ? "unsigned short" : ( sizeof( ATLAS_T ) == 4 ? "unsigned int" : "unsigned long" ) ),
                              (int)GROUP, BIG ? "true" : "false" );
    vrc_internal_note_kernel_fn( (const void*)&vrc_k_raycast< DDA, CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG >,
                                 (int)VRC_WG_THREADS, 0 );
    hipLaunchKernelGGL( ( vrc_k_raycast< DDA, CLAMP, COUNT, FIXED, MODE, ATLAS_T, GROUP, BIG > ),
                        dim3( ( vrc_schedule_slots( tilesX, tilesY ) + VRC_WAVES_PER_WG - 1u ) / VRC_WAVES_PER_WG ),
                        dim3( VRC_WG_THREADS ), 0, stream, a.frame, a.nodes, a.gridTable,
                        (const ATLAS_T*)a.atlas, a.lut, a.classifier, a.pixelBuffer,
                        a.sampleCounter, a.tileOrder, tilesX, nTiles );
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * Depth split (VRC_OPT_DEPTH_SPLIT): the same integrator with TWO waves per tile -- wave h of a tile marches the
 * bricks whose segments start in the near (h = 0) or far (h = 1) half of each ray's interval inside the brick grid
 * (vrc_ray_grid_dda, partDir < 0) -- composited with the
 * `over` operator through LDS.  For launches that cannot fill the machine (a rank's share of a sort-first frame):
 * the time of such a launch is the longest ray's dependent chain of samples, and this halves it.  Every brick
 * is marched whole by exactly one of the two waves, so the samples are the reference's; compositing the far half
 * from zero and blending it in afterwards regroups the float additions (differences ~1e-7, inside E0).  Early ray
 * termination would need the near half's opacity inside the far half: the launcher takes this kernel only for frames
 * whose classified table cannot reach the threshold (vrc_api.hip), and only for the first pass of a frame.
 * Workgroup = 8 waves: waves 0-3 the near halves of the unit's four tiles, waves 4-7 the far halves.
 * ---------------------------------------------------------------------------------------- */
template < bool COUNT, int GROUP >
__global__ __launch_bounds__( 512, GROUP > 8 ? 2 : 4 ) void vrc_k_raycast_split(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes, const int32_t* __restrict__ gridTable,
    const uint8_t* __restrict__ atlas, const vrc_f4* __restrict__ lutGlobal, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles, const int partDir )
{
    __shared__ vrc_f4 lut[VRC_TFP_ENTRIES];
    __shared__ vrc_f4 farColor[4][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, sub = wave & 3u;
    const int half = (int)( wave >> 2 );
    if( tid < 256u )
        lut[tid] = lutGlobal[tid];
    if( tid < VRC_TFP_ENTRIES - 256u )
        lut[256u + tid] = lutGlobal[256u + tid];
#if defined( VRC_ADDR_TABLES )
    if( tid < 256u )
    {
        const vrc_lay lay = vrc_make_lay( f.sbx, f.sby );
        vrc_addr_tab[tid] = vrc_lay_x( lay, tid );
        vrc_addr_tab[256u + tid] = vrc_lay_y( lay, tid );
        vrc_addr_tab[512u + tid] = vrc_lay_z( lay, tid );
    }
#endif
    __syncthreads();
    const uint32_t tilesY = nTiles / tilesX;
    const uint32_t slotIndex = blockIdx.x * 4u + sub;
    const uint32_t tile = slotIndex < vrc_schedule_slots( tilesX, tilesY )
                              ? vrc_slot_tile( tileOrder, slotIndex, tilesX, tilesY )
                              : VRC_NO_TILE;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
    const uint32_t ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
    const uint32_t px = tx * 8u + lx, py = ty * 8u + ly;
    const bool inFrame = tile != VRC_NO_TILE && px < f.width && py < f.height;

    uint32_t nSamples = 0;
    vrc_f4 color = { 0.f, 0.f, 0.f, 0.f };
    bool hit = false;
    if( inFrame )
    {
        const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
        hit = r.hit;
        if( hit )
            vrc_ray_grid_dda< false, COUNT, true, VRC_MODE_TABLE, uint8_t, GROUP, false >(
                f, r, nodes, gridTable, atlas, lut, cls, color, nSamples, half, 2, partDir );
    }
    if( half == 1 )
        farColor[sub][lane] = color;
    __syncthreads(); /* every wave of the workgroup gets here: nothing above returns */
    if( half == 0 && inFrame )
    {
        if( hit )
        {
            /* front-to-back `over`: near + (1 - near.alpha) * far; both hold opacity-weighted colour */
            const vrc_f4 farC = farColor[sub][lane];
            const float t = 1.0f - color.w;
            color.x += farC.x * t;
            color.y += farC.y * t;
            color.z += farC.z * t;
            color.w += farC.w * t;
        }
        pixelBuffer[py * f.width + px] = color; /* the folded clear: a missed pixel is written as 0 */
    }
    if( COUNT )
    {
        unsigned long long s = nSamples;
#pragma unroll
        for( int off = 32; off > 0; off >>= 1 )
            s += __shfl_down( s, off, 64 );
        if( lane == 0 && s != 0 )
            atomicAdd( sampleCounter, s );
    }
}

template < bool COUNT, int GROUP >
static hipError_t launch_split( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + 7u ) / 8u, tilesY = ( a.frame.height + 7u ) / 8u;
    if( tilesX * tilesY == 0 )
        return hipSuccess;
    vrc_internal_note_kernel( "vrc_k_raycast_split<%s,%d>", COUNT ? "true" : "false", (int)GROUP );
    hipLaunchKernelGGL( ( vrc_k_raycast_split< COUNT, GROUP > ), dim3( vrc_schedule_slots( tilesX, tilesY ) / 4u ),
                        dim3( 512 ), 0, stream, a.frame, a.nodes, a.gridTable, (const uint8_t*)a.atlas, a.lut,
                        a.classifier, a.pixelBuffer, a.sampleCounter, a.tileOrder, tilesX, tilesX * tilesY,
                        -1 /* halves of every ray's own interval: measured better balanced than grid slabs from 4 ranks
                              down */ );
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------
 * Ray compaction (VRC_OPT_ERT_COMPACTION = P): the march in P launches, one per slab of the brick grid along the
 * view axis (vrc_ray_grid_dda's `part`).  The first launch is the tile-scheduled kernel above; at its
 * end every wave takes a ballot of the rays that early termination (Renderer.cu:219-226) has not ended, reserves
 * popcount(ballot) entries of a ray list with one atomic and each live lane writes its pixel at the position
 * mbcnt(ballot) gives it.  The following launches take their rays from that list, 64 consecutive entries per wave:
 * full waves of live rays instead of tiles in which most lanes have finished.  A ray's bricks are marched in the same
 * order with the same arithmetic as in one launch and its colour travels through the pixel buffer as float32, so the
 * frame is bit-identical to the single launch; rays that end early never reach the later, sparser launches.
 * What it costs: the dead lanes it removes cost the vector ALUs issue slots but almost no L1 tag look-ups (a quad
 * without live lanes is skipped), while packed survivors from different tiles share fewer cache lines per quad
 *  -- so it is an option, measured in DESIGN.md, and off by default.
 * ---------------------------------------------------------------------------------------- */
template < bool COUNT, bool LISTED >
__global__ __launch_bounds__( VRC_WG_THREADS, 4 ) void vrc_k_raycast_part(
    const vrc_frame f, const vrc_dev_node* __restrict__ nodes, const int32_t* __restrict__ gridTable,
    const uint8_t* __restrict__ atlas, const vrc_f4* __restrict__ lutGlobal, const vrc_classifier cls,
    vrc_f4* __restrict__ pixelBuffer, unsigned long long* __restrict__ sampleCounter,
    const uint32_t* __restrict__ tileOrder, const uint32_t tilesX, const uint32_t nTiles, const int part,
    const int parts, const int partDir, const uint32_t* __restrict__ listIn, const uint32_t* __restrict__ countIn,
    uint32_t* __restrict__ listOut, uint32_t* __restrict__ countOut )
{
    __shared__ vrc_f4 lut[VRC_TFP_ENTRIES];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t nRays = 0;
    if( LISTED )
    {
        nRays = *countIn;
        if( blockIdx.x * VRC_WG_THREADS >= nRays ) /* the whole workgroup: the grid is sized for every pixel */
            return;
    }
#pragma unroll
    for( uint32_t i = 0; i < ( 256u + VRC_WG_THREADS - 1u ) / VRC_WG_THREADS; ++i )
        if( tid + i * VRC_WG_THREADS < 256u )
            lut[tid + i * VRC_WG_THREADS] = lutGlobal[tid + i * VRC_WG_THREADS];
    if( tid < VRC_TFP_ENTRIES - 256u )
        lut[256u + tid] = lutGlobal[256u + tid];
#if defined( VRC_ADDR_TABLES )
    {
        const vrc_lay lay = vrc_make_lay( f.sbx, f.sby );
#pragma unroll
        for( uint32_t i = 0; i < ( 256u + VRC_WG_THREADS - 1u ) / VRC_WG_THREADS; ++i )
        {
            const uint32_t u = tid + i * VRC_WG_THREADS;
            if( u < 256u )
            {
                vrc_addr_tab[u] = vrc_lay_x( lay, u );
                vrc_addr_tab[256u + u] = vrc_lay_y( lay, u );
                vrc_addr_tab[512u + u] = vrc_lay_z( lay, u );
            }
        }
    }
#endif
    __syncthreads();
    /* from here on the waves of the workgroup are independent */
    uint32_t px = 0, py = 0;
    bool valid;
    if( LISTED )
    {
        const uint32_t gid = blockIdx.x * VRC_WG_THREADS + tid;
        valid = gid < nRays;
        const uint32_t packed = valid ? listIn[gid] : 0u;
        px = packed & 0xFFFFu;
        py = packed >> 16;
    }
    else
    {
        const uint32_t slotIndex = blockIdx.x * VRC_WAVES_PER_WG + ( tid >> 6 );
        const uint32_t tilesY = nTiles / tilesX;
        if( slotIndex >= vrc_schedule_slots( tilesX, tilesY ) )
            return;
        const uint32_t tile = vrc_slot_tile( tileOrder, slotIndex, tilesX, tilesY );
        if( tile == VRC_NO_TILE )
            return;
        const uint32_t lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
        const uint32_t ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
        px = ( tile % tilesX ) * 8u + lx;
        py = ( tile / tilesX ) * 8u + ly;
        valid = px < f.width && py < f.height;
    }

    uint32_t nSamples = 0;
    bool alive = false;
    if( valid )
    {
        const vrc_ray r = vrc_setup_ray( f, px, f.rowMap ? f.rowMap[py] : py );
        const uint32_t pixelPos = py * f.width + px;
        const vrc_f4 zero = { 0.f, 0.f, 0.f, 0.f };
        if( !r.hit )
        {
            if( !LISTED && f.clearFirst )
                pixelBuffer[pixelPos] = zero;
        }
        else
        {
            /* the first launch starts like vrc_pixel_grid_dda; the later ones go on from what the one before
             * left in the pixel buffer */
            vrc_f4 color = ( LISTED || !f.clearFirst ) ? pixelBuffer[pixelPos] : zero;
            if( !( color.w > VRC_EARLY_EXIT ) )
            {
                alive = vrc_ray_grid_dda< false, COUNT, true, VRC_MODE_TABLE, uint8_t, VRC_GROUP, false >(
                    f, r, nodes, gridTable, atlas, lut, cls, color, nSamples, part, parts, partDir );
                pixelBuffer[pixelPos] = color;
            }
        }
    }
    if( listOut )
    {
        /* wave-level compaction: one atomic per wave, order inside the wave kept (Morton order of a tile) */
        const uint64_t live = __builtin_amdgcn_ballot_w64( alive );
        if( live != 0ull )
        {
            uint32_t base = 0;
            const uint32_t first = (uint32_t)__builtin_ctzll( live );
            if( lane == first )
                base = atomicAdd( countOut, (uint32_t)__builtin_popcountll( live ) );
            base = (uint32_t)__builtin_amdgcn_readlane( (int)base, (int)first );
            if( alive )
            {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi( (uint32_t)( live >> 32 ),
                                                                  __builtin_amdgcn_mbcnt_lo( (uint32_t)live, 0u ) );
                listOut[base + below] = ( py << 16 ) | px;
            }
        }
    }
    if( COUNT )
    {
        unsigned long long s = nSamples;
#pragma unroll
        for( int off = 32; off > 0; off >>= 1 )
            s += __shfl_down( s, off, 64 );
        if( lane == 0 && s != 0 )
            atomicAdd( sampleCounter, s );
    }
}

template < bool COUNT >
static hipError_t launch_parts( const vrc_raycast_args& a, hipStream_t stream )
{
    const uint32_t tilesX = ( a.frame.width + 7u ) / 8u, tilesY = ( a.frame.height + 7u ) / 8u;
    if( tilesX * tilesY == 0 )
        return hipSuccess;
    const int parts = a.ertParts;
    const int partDir = vrc_part_dir( a.frame );
    /* rayList: counts[VRC_MAX_ERT_PARTS] | list 0 [width * height] | list 1 [width * height] */
    uint32_t* const counts = a.rayList;
    uint32_t* const lists[2] = { a.rayList + VRC_MAX_ERT_PARTS,
                                 a.rayList + VRC_MAX_ERT_PARTS + (size_t)a.frame.width * a.frame.height };
    hipError_t e = hipMemsetAsync( counts, 0, VRC_MAX_ERT_PARTS * sizeof( uint32_t ), stream );
    if( e != hipSuccess )
        return e;
    hipLaunchKernelGGL( ( vrc_k_raycast_part< COUNT, false > ),
                        dim3( ( vrc_schedule_slots( tilesX, tilesY ) + VRC_WAVES_PER_WG - 1u ) / VRC_WAVES_PER_WG ),
                        dim3( VRC_WG_THREADS ), 0, stream, a.frame, a.nodes, a.gridTable, (const uint8_t*)a.atlas, a.lut,
                        a.classifier, a.pixelBuffer, a.sampleCounter, a.tileOrder, tilesX, tilesX * tilesY, 0, parts, partDir,
                        (const uint32_t*)nullptr, (const uint32_t*)nullptr, lists[0], counts );
    const uint32_t allRays = ( a.frame.width * a.frame.height + VRC_WG_THREADS - 1u ) / VRC_WG_THREADS;
    for( int p = 1; p < parts; ++p )
        hipLaunchKernelGGL( ( vrc_k_raycast_part< COUNT, true > ), dim3( allRays ), dim3( VRC_WG_THREADS ), 0, stream,
                            a.frame, a.nodes, a.gridTable, (const uint8_t*)a.atlas, a.lut, a.classifier, a.pixelBuffer,
                            a.sampleCounter, a.tileOrder, tilesX, tilesX * tilesY, p, parts, partDir,
                            (const uint32_t*)lists[( p - 1 ) & 1], (const uint32_t*)( counts + ( p - 1 ) ),
                            p + 1 < parts ? lists[p & 1] : (uint32_t*)nullptr, counts + p );
    return hipGetLastError();
}

/* per-sample classification modes: trilinear or point, u8 or u16 voxels.  Point sampling of 16-bit voxels with
 * overlap (no clamped sampler) takes the fixed-point grouped march (vrc_march_brick) */
template < int MODE, typename ATLAS_T >
static hipError_t launch_classify( const vrc_raycast_args& a, bool count, hipStream_t stream )
{
    if constexpr( MODE == VRC_MODE_POINT && sizeof( ATLAS_T ) == 2 )
        if( a.fixedStepping && !a.clamp )
        {
            if( a.greyTable )
            {
                if( a.gridDda )
                    return count ? launch_variant< true, false, true, true, VRC_MODE_POINT_GREY, ATLAS_T >( a, stream )
                                 : launch_variant< true, false, false, true, VRC_MODE_POINT_GREY, ATLAS_T >( a, stream );
                return count ? launch_variant< false, false, true, true, VRC_MODE_POINT_GREY, ATLAS_T >( a, stream )
                             : launch_variant< false, false, false, true, VRC_MODE_POINT_GREY, ATLAS_T >( a, stream );
            }
            if( a.gridDda )
                return count ? launch_variant< true, false, true, true, MODE, ATLAS_T >( a, stream )
                             : launch_variant< true, false, false, true, MODE, ATLAS_T >( a, stream );
            return count ? launch_variant< false, false, true, true, MODE, ATLAS_T >( a, stream )
                         : launch_variant< false, false, false, true, MODE, ATLAS_T >( a, stream );
        }
    const int key = ( a.gridDda ? 4 : 0 ) | ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 );
    switch( key )
    {
    case 0: return launch_variant< false, false, false, false, MODE, ATLAS_T >( a, stream );
    case 1: return launch_variant< false, false, true, false, MODE, ATLAS_T >( a, stream );
    case 2: return launch_variant< false, true, false, false, MODE, ATLAS_T >( a, stream );
    case 3: return launch_variant< false, true, true, false, MODE, ATLAS_T >( a, stream );
    case 4: return launch_variant< true, false, false, false, MODE, ATLAS_T >( a, stream );
    case 5: return launch_variant< true, false, true, false, MODE, ATLAS_T >( a, stream );
    case 6: return launch_variant< true, true, false, false, MODE, ATLAS_T >( a, stream );
    default: return launch_variant< true, true, true, false, MODE, ATLAS_T >( a, stream );
    }
}

/* atlases of more than 2^32 voxels: 64-bit slot bases (BIG), groups of 8; float stepping except for point sampling of
 * 16-bit voxels with overlap, which takes the fixed-point grouped march as in launch_classify */
template < int MODE, typename ATLAS_T >
static hipError_t launch_big( const vrc_raycast_args& a, bool count, hipStream_t stream )
{
    if constexpr( MODE == VRC_MODE_POINT && sizeof( ATLAS_T ) == 2 )
        if( a.fixedStepping && !a.clamp )
        {
#define VRC_BIG_FIXED( M )                                                                                         \
    ( a.gridDda ? ( count ? launch_variant< true, false, true, true, M, ATLAS_T, VRC_GROUP, true >( a, stream )     \
                          : launch_variant< true, false, false, true, M, ATLAS_T, VRC_GROUP, true >( a, stream ) )  \
                : ( count ? launch_variant< false, false, true, true, M, ATLAS_T, VRC_GROUP, true >( a, stream )    \
                          : launch_variant< false, false, false, true, M, ATLAS_T, VRC_GROUP, true >( a, stream ) ) )
            return a.greyTable ? VRC_BIG_FIXED( VRC_MODE_POINT_GREY ) : VRC_BIG_FIXED( VRC_MODE_POINT );
#undef VRC_BIG_FIXED
        }
    const int key = ( a.gridDda ? 4 : 0 ) | ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 );
    switch( key )
    {
    case 0: return launch_variant< false, false, false, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 1: return launch_variant< false, false, true, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 2: return launch_variant< false, true, false, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 3: return launch_variant< false, true, true, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 4: return launch_variant< true, false, false, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 5: return launch_variant< true, false, true, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    case 6: return launch_variant< true, true, false, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    default: return launch_variant< true, true, true, false, MODE, ATLAS_T, VRC_GROUP, true >( a, stream );
    }
}

/* trilinear through the tap-packed atlas (a.atlas = the packed atlas; the host offers it for 8-bit bricks with
 * or 16-bit bricks with overlap >= 1, slots of at most 248 voxels a side, VRC_OPT_TF_FRAC_BITS = 8);
 * WIDE: a packed atlas of more than 4 GiB or of an atlas of more than 2^32 voxels */
template < int MODE, bool WIDE, typename TAG >
static hipError_t launch_packed( const vrc_raycast_args& a, bool count, hipStream_t stream )
{
    /* TAG: uint32_t = the packed atlas of 8-bit voxels (16-bit texels, groups of VRC_PGROUP), uint64_t = of 16-bit
     * voxels (32-bit texels: twice the registers per sample in flight, groups of VRC_PGROUP16) */
    constexpr int G = sizeof( TAG ) == 8 ? VRC_PGROUP16 : VRC_PGROUP;
    if( a.gridDda )
        return count ? launch_variant< true, false, true, true, MODE, TAG, G, WIDE >( a, stream )
                     : launch_variant< true, false, false, true, MODE, TAG, G, WIDE >( a, stream );
    return count ? launch_variant< false, false, true, true, MODE, TAG, G, WIDE >( a, stream )
                 : launch_variant< false, false, false, true, MODE, TAG, G, WIDE >( a, stream );
}
template < bool WIDE, typename TAG >
static hipError_t launch_packed( const vrc_raycast_args& a, bool count, hipStream_t stream )
{
    return a.greyTable ? launch_packed< VRC_MODE_PACKED_GREY, WIDE, TAG >( a, count, stream )
                       : launch_packed< VRC_MODE_PACKED, WIDE, TAG >( a, count, stream );
}

hipError_t vrc_launch_raycast( const vrc_raycast_args& a, hipStream_t stream )
{
    const bool count = a.sampleCounter != nullptr;
    if( a.packed )
    {
        if( !a.linear || ( a.elemBytes != 1 && a.elemBytes != 2 ) || a.clamp || ( a.bigAtlas && !a.packedWide ) )
            return hipErrorInvalidValue;
        if( a.elemBytes == 2 )
            return a.packedWide ? launch_packed< true, uint64_t >( a, count, stream ) : launch_packed< false, uint64_t >( a, count, stream );
        return a.packedWide ? launch_packed< true, uint32_t >( a, count, stream ) : launch_packed< false, uint32_t >( a, count, stream );
    }
    if( a.bigAtlas )
    {
        if( a.elemBytes == 2 )
            return a.linear ? launch_big< VRC_MODE_TRILINEAR, uint16_t >( a, count, stream )
                            : launch_big< VRC_MODE_POINT, uint16_t >( a, count, stream );
        if( a.elemBytes != 1 )
            return hipErrorInvalidValue;
        return a.linear ? launch_big< VRC_MODE_TRILINEAR, uint8_t >( a, count, stream )
                        : launch_big< VRC_MODE_TABLE, uint8_t >( a, count, stream );
    }
    if( a.elemBytes == 2 )
        return a.linear ? launch_classify< VRC_MODE_TRILINEAR, uint16_t >( a, count, stream )
                        : launch_classify< VRC_MODE_POINT, uint16_t >( a, count, stream );
    if( a.elemBytes != 1 )
        return hipErrorInvalidValue;
    if( a.linear )
        return launch_classify< VRC_MODE_TRILINEAR, uint8_t >( a, count, stream );
    /* the clamped sampler (overlap 0) always uses the float position chain */
    const bool fixed = a.fixedStepping && !a.clamp;
    const int key = ( fixed ? 8 : 0 ) | ( a.gridDda ? 4 : 0 ) | ( a.clamp ? 2 : 0 ) | ( count ? 1 : 0 );
    /* launches of at most ~1.5 waves per SIMD slot (256 CUs x 4 SIMDs x 4 waves) are latency-bound:
     * measured 0.153 -> 0.130 ms for an eighth of the C2 frame, 0.538 -> 0.562 ms for the whole */
    const uint32_t nTilesLaunch = ( ( a.frame.width + VRC_TILE_W - 1 ) / VRC_TILE_W ) *
                                  ( ( a.frame.height + VRC_TILE_H - 1 ) / VRC_TILE_H );
#ifndef VRC_SMALL_GROUP
#define VRC_SMALL_GROUP 24 /* samples in flight per lane in launches too small to fill the GPU (latency-bound: 16 -> 24: -5 % on a rank's share of an 8-rank frame, -4 % of a 4-rank frame; 32: -7 % / -3 %) */
#endif
#ifndef VRC_SMALL_LAUNCH_TILES
#define VRC_SMALL_LAUNCH_TILES 6144u /* (8192 = half a 1024^2 frame: one frame alone 0.303 -> 0.287 ms, three in flight 4428 -> 4365 frames/s: not taken) */
#endif
    const bool smallLaunch = nTilesLaunch <= VRC_SMALL_LAUNCH_TILES;
    if( a.ertParts > 1 && ( key == 12 || key == 13 ) && VRC_TILE_W == 8u )
        return count ? launch_parts< true >( a, stream ) : launch_parts< false >( a, stream );
    if( a.depthSplit && ( key == 12 || key == 13 ) && VRC_TILE_W == 8u )
    {
#ifndef VRC_SPLIT_GROUP
#define VRC_SPLIT_GROUP 8
#endif
        return count ? launch_split< true, VRC_SPLIT_GROUP >( a, stream ) : launch_split< false, VRC_SPLIT_GROUP >( a, stream );
    }
    switch( key )
    {
    case 0: return launch_variant< false, false, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 1: return launch_variant< false, false, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 2: return launch_variant< false, true, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 3: return launch_variant< false, true, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 4: return launch_variant< true, false, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 5: return launch_variant< true, false, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 6: return launch_variant< true, true, false, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 7: return launch_variant< true, true, true, false, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 8:
        if( a.greyTable )
            return launch_variant< false, false, false, true, VRC_MODE_GREY, uint8_t, VRC_GREY_GROUP >( a, stream );
        return launch_variant< false, false, false, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 9:
        if( a.greyTable )
            return launch_variant< false, false, true, true, VRC_MODE_GREY, uint8_t, VRC_GREY_GROUP >( a, stream );
        return launch_variant< false, false, true, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    case 12:
        if( a.greyTable )
            return smallLaunch ? launch_variant< true, false, false, true, VRC_MODE_GREY, uint8_t, VRC_SMALL_GROUP >( a, stream )
                               : launch_variant< true, false, false, true, VRC_MODE_GREY, uint8_t, VRC_GREY_GROUP >( a, stream );
        if( smallLaunch )
            return launch_variant< true, false, false, true, VRC_MODE_TABLE, uint8_t, VRC_SMALL_GROUP >( a, stream );
        return launch_variant< true, false, false, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    default:
        if( a.greyTable )
            return smallLaunch ? launch_variant< true, false, true, true, VRC_MODE_GREY, uint8_t, VRC_SMALL_GROUP >( a, stream )
                               : launch_variant< true, false, true, true, VRC_MODE_GREY, uint8_t, VRC_GREY_GROUP >( a, stream );
        if( smallLaunch )
            return launch_variant< true, false, true, true, VRC_MODE_TABLE, uint8_t, VRC_SMALL_GROUP >( a, stream );
        return launch_variant< true, false, true, true, VRC_MODE_TABLE, uint8_t >( a, stream );
    }
}
