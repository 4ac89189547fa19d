/*
 * harness.cpp -- host build (g++) of the per-ray logic in libre_amd/csrc/vrc_core.h and the
 * table builder in vrc_tables.h, for CPU-side unit tests (optionally under ASan/UBSan) in a
 * container without a GPU.  TEST INFRASTRUCTURE ONLY: nothing in the product loads this; the
 * product path is the gfx950 kernel in vrc_kernels.hip, which includes the same headers.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../libre_amd/csrc/vrc_tables.h"

static int g_parts = 0;

static int render_impl( const uint8_t* atlasRowMajor, const uint32_t atlasDim[3],
                        const uint32_t slotDim[3], float* pixelBuffer, uint32_t W,
                        uint32_t H, const float* planes, uint32_t nPlanes, const float* tf,
                        const vrc_view_data* view, uint32_t nNodes,
                        const vrc_node_data* nodes, const vrc_render_data* render,
                        int fracBits, int kernel, int pixelOffX, int pixelOffY,
                        uint64_t* samplesOut, int* gridOkOut, int voxelBytes, int variant,
                        bool rayLod, float lodSse, float lodWorldPerPixel )
{
    if( voxelBytes != 1 && voxelBytes != 2 )
        return 3;
    vrc_atlas_geom geom;
    for( int a = 0; a < 3; ++a )
    {
        if( atlasDim[a] % 8u || slotDim[a] % 8u )
            return 1;
        geom.atlasDim[a] = atlasDim[a];
        geom.slotDim[a] = slotDim[a];
    }
    vrc_layout lay;
    for( int a = 0; a < 3; ++a )
    {
        geom.slots[a] = atlasDim[a] / slotDim[a];
        lay.slots[a] = geom.slots[a];
        lay.slotDim[a] = slotDim[a];
    }

    /* micro-blocked copy of the atlas, as the upload kernel lays it out */
    const size_t nVoxels = (size_t)atlasDim[0] * atlasDim[1] * atlasDim[2];
    std::vector< uint8_t > atlas( voxelBytes == 1 ? nVoxels : 0 );
    std::vector< uint16_t > atlas16( voxelBytes == 2 ? nVoxels : 0 );
    for( uint32_t z = 0; z < atlasDim[2]; ++z )
        for( uint32_t y = 0; y < atlasDim[1]; ++y )
            for( uint32_t x = 0; x < atlasDim[0]; ++x )
            {
                const size_t src = ( (size_t)z * atlasDim[1] + y ) * atlasDim[0] + x;
                if( voxelBytes == 1 )
                    atlas[vrc_atlas_index( lay, x, y, z )] = atlasRowMajor[src];
                else
                    atlas16[vrc_atlas_index( lay, x, y, z )] =
                        reinterpret_cast< const uint16_t* >( atlasRowMajor )[src];
            }

    vrc_lut_params lp;
    lp.rangeMin = render->dataSourceRange[0];
    lp.rangeMax = render->dataSourceRange[1];
    lp.alphaCorrection = (float)render->maxSamplesPerRay / (float)render->samplesPerRay;
    lp.fracBits = fracBits;
    /* one classified table per level (per-ray LOD: exponent doubled per level), as vrc_render builds them */
    const uint32_t lutLevels = rayLod ? (uint32_t)VRC_MAX_LOD_LEVELS : 1u;
    std::vector< vrc_f4 > lut( lutLevels * VRC_LUT_ENTRIES );
    for( uint32_t j = 0; j < lutLevels; ++j )
    {
        vrc_lut_params lj = lp;
        lj.alphaCorrection = lp.alphaCorrection * (float)( 1u << j );
        for( uint32_t d = 0; d < 256; ++d )
            lut[j * VRC_LUT_ENTRIES + d] = vrc_lut_entry( tf, d, lj );
        lut[j * VRC_LUT_ENTRIES + 256] = vrc_f4{ 0.f, 0.f, 0.f, 0.f };
    }

    vrc_host_tables t;
    vrc_build_tables( geom, nodes, nNodes, t );
    if( rayLod )
        vrc_build_lod_tables( geom, nodes, nNodes, t );
    if( gridOkOut )
        *gridOkOut = ( rayLod ? t.lodOk : t.gridOk ) ? 1 : 0;
    if( rayLod && ( !t.lodOk || variant != 0 ) )
        return 2;
    float pl[6][4];
    std::memset( pl, 0, sizeof( pl ) );
    for( uint32_t i = 0; i < nPlanes && i < 6; ++i )
        for( int k = 0; k < 4; ++k )
            pl[i][k] = planes[i * 4 + k];
    vrc_frame f;
    std::memset( &f, 0, sizeof( f ) );
    const float centre = variant == 1 ? 0.5f : 0.0f; /* glRaycaster: gl_FragCoord */
    vrc_fill_frame( f, *view, *render, geom, t.g, pl, nPlanes, nNodes, W, H, (float)pixelOffX + centre,
                    (float)pixelOffY + centre );
    f.variant = variant == 1 ? VRC_VARIANT_GL : VRC_VARIANT_CUDA;
    f.samplesPerPixel = ( variant == 1 && render->samplesPerPixel > 1u ) ? render->samplesPerPixel : 1u;
    if( rayLod )
    {
        f.lodLevels = t.lodLevels;
        f.lodBase = (float)( t.finestVoxelWorld / ( (double)lodSse * (double)lodWorldPerPixel ) );
    }

    /* kernel: 1 reference order, 2 grid DDA, 3/4 the same with fixed-point stepping (u8 only),
     * 5/6 the same with the trilinear filter, 7/8 point sampling with per-sample
     * classification (the only point-sampling form for 16-bit voxels), 9/10 the trilinear filter through the
     * tap-packed atlas (u8 or u16, overlap >= 1, fracBits 8), 11/12 the same with the (grey, alpha) colours of a grey
     * transfer function */
    const bool dda = !rayLod && ( kernel == 2 || kernel == 4 || kernel == 6 || kernel == 8 || kernel == 10 || kernel == 12 );
    const bool packedKernel = kernel >= 9 && kernel <= 12;
    if( packedKernel && ( ( voxelBytes != 1 && voxelBytes != 2 ) || t.clamp || fracBits != 8 ) )
        return 5;
    /* the tap-packed atlas as vrc_k_pack_slots writes it (vrc_core.h): texels of twice the voxel's bytes in rows of 9, the
     * z neighbour and the ninth column's copy clamped at the slot's last voxel (+ 2 words: a pair is read as 4 / 8 bytes) */
    const uint32_t texelBytes = VRC_PK_TEXEL( (uint32_t)voxelBytes );
    std::vector< uint32_t > packed( packedKernel ? (size_t)( vrc_packed_elems( nVoxels ) * texelBytes / 4u + 2u ) : 0 );
    if( packedKernel )
    {
        const uint32_t sbx = slotDim[0] / 8u, sby = slotDim[1] / 8u;
        for( uint32_t k = 0; k < geom.slots[2]; ++k )
            for( uint32_t j = 0; j < geom.slots[1]; ++j )
                for( uint32_t i = 0; i < geom.slots[0]; ++i )
                {
                    const uint64_t base = vrc_slot_base( lay, i, j, k ); /* of the atlas's slot */
                    for( uint32_t z = 0; z < slotDim[2]; ++z )
                        for( uint32_t y = 0; y < slotDim[1]; ++y )
                            for( uint32_t bx = 0; bx < sbx; ++bx )
                                for( uint32_t ix = 0; ix < VRC_PK_ROW; ++ix )
                                {
                                    uint32_t x = bx * 8u + ix;
                                    x = x < slotDim[0] ? x : slotDim[0] - 1u;
                                    const uint32_t z1 = z + 1u < slotDim[2] ? z + 1u : z;
                                    const uint64_t e0 = base + vrc_slot_local_index( x, y, z, sbx, sby ),
                                                   e1 = base + vrc_slot_local_index( x, y, z1, sbx, sby );
                                    const uint64_t o = vrc_packed_elems( base ) + vrc_packed_local_index( bx * 8u, y, z, sbx, sby ) + ix;
                                    if( voxelBytes == 1 )
                                        reinterpret_cast< uint16_t* >( packed.data() )[o] = vrc_pack_taps< uint16_t >( atlas[e0], atlas[e1] );
                                    else
                                        packed[o] = vrc_pack_taps< uint32_t >( atlas16[e0], atlas16[e1] );
                                }
                }
    }
    const int mode = ( kernel == 5 || kernel == 6 ) ? VRC_MODE_TRILINEAR
                     : ( ( kernel == 7 || kernel == 8 ) ? VRC_MODE_POINT : VRC_MODE_TABLE );
    const bool fixed = ( kernel == 3 || kernel == 4 ) && !t.clamp;
    if( dda && !t.gridOk )
        return 2;
    if( voxelBytes == 2 && mode == VRC_MODE_TABLE && !packedKernel )
        return 4; /* the classified table indexes 8-bit voxels only */
    const vrc_classifier cls = vrc_make_classifier( lp );
    std::vector< vrc_f4 > tfp( VRC_TFP_ENTRIES );
    for( uint32_t k = 0; k < VRC_TFP_ENTRIES; ++k )
    {
        const uint32_t i = k == 0 ? 0u : ( k - 1u > 255u ? 255u : k - 1u );
        tfp[k] = vrc_f4{ tf[i * 4], tf[i * 4 + 1], tf[i * 4 + 2], tf[i * 4 + 3] };
    }
    const vrc_f4* table = mode != VRC_MODE_TABLE ? tfp.data() : lut.data();
    std::vector< vrc_f4 > cls8( VRC_CLS8_ENTRIES );
    if( packedKernel )
    {
        for( uint32_t k = 0; k < VRC_CLS8_ENTRIES; ++k )
            cls8[k] = vrc_cls8_entry( tfp.data(), k, kernel >= 11 );
        table = cls8.data();
    }
    uint64_t total = 0;
    vrc_f4* pb = reinterpret_cast< vrc_f4* >( pixelBuffer );
    for( uint32_t py = 0; py < H; ++py )
        for( uint32_t px = 0; px < W; ++px )
        {
            uint32_t n = 0;
#define ARGS_DDA( A ) f, t.nodes.data(), t.grid.data(), A, table, cls, pb, px, py, n
#define ARGS_REF( A ) f, t.nodes.data(), A, table, cls, pb, px, py, n
#define CLASSIFY( MODE, T, A )                                                                       \
    {                                                                                                \
        if( dda && t.clamp ) vrc_pixel_grid_dda< true, true, false, MODE, T >( ARGS_DDA( A ) );      \
        else if( dda ) vrc_pixel_grid_dda< false, true, false, MODE, T >( ARGS_DDA( A ) );           \
        else if( t.clamp ) vrc_pixel_reference_order< true, true, false, MODE, T >( ARGS_REF( A ) ); \
        else vrc_pixel_reference_order< false, true, false, MODE, T >( ARGS_REF( A ) );              \
    }
#define RAYLOD( FIXED, MODE, T, A )                                                                     \
    {                                                                                                \
        if( t.clamp ) vrc_pixel_ray_lod< true, true, false, MODE, T >( ARGS_DDA( A ) );              \
        else vrc_pixel_ray_lod< false, true, FIXED, MODE, T >( ARGS_DDA( A ) );                      \
    }
#define PACKED( TAG, G )                                                                                                                  \
    {                                                                                                                                     \
        const TAG* const pk = reinterpret_cast< const TAG* >( packed.data() );                                                            \
        if( rayLod && kernel >= 11 ) vrc_pixel_ray_lod< false, true, true, VRC_MODE_PACKED_GREY, TAG, G >( ARGS_DDA( pk ) );              \
        else if( rayLod ) vrc_pixel_ray_lod< false, true, true, VRC_MODE_PACKED, TAG, G >( ARGS_DDA( pk ) );                              \
        else if( kernel == 9 ) vrc_pixel_reference_order< false, true, true, VRC_MODE_PACKED, TAG, G >( ARGS_REF( pk ) );                 \
        else if( kernel == 10 ) vrc_pixel_grid_dda< false, true, true, VRC_MODE_PACKED, TAG, G >( ARGS_DDA( pk ) );                       \
        else if( kernel == 11 ) vrc_pixel_reference_order< false, true, true, VRC_MODE_PACKED_GREY, TAG, G >( ARGS_REF( pk ) );           \
        else vrc_pixel_grid_dda< false, true, true, VRC_MODE_PACKED_GREY, TAG, G >( ARGS_DDA( pk ) );                                     \
    }
            if( packedKernel )
            {
                /* the tap-packed march (9 / 10: reference order / grid walk; 11 / 12: grey colours; per-ray LOD around it:
                 * 9 or 11); the tag says which packed atlas: uint32_t of 8-bit voxels, uint64_t of 16-bit voxels */
                if( voxelBytes == 1 ) PACKED( uint32_t, VRC_PGROUP )
                else PACKED( uint64_t, VRC_PGROUP16 )
            }
#undef PACKED
            else if( rayLod )
            {
                if( mode == VRC_MODE_TRILINEAR && voxelBytes == 1 ) RAYLOD( false, VRC_MODE_TRILINEAR, uint8_t, atlas.data() )
                else if( mode == VRC_MODE_TRILINEAR ) RAYLOD( false, VRC_MODE_TRILINEAR, uint16_t, atlas16.data() )
                else if( mode == VRC_MODE_POINT && voxelBytes == 1 ) RAYLOD( false, VRC_MODE_POINT, uint8_t, atlas.data() )
                else if( mode == VRC_MODE_POINT ) RAYLOD( false, VRC_MODE_POINT, uint16_t, atlas16.data() )
                else if( fixed ) RAYLOD( true, VRC_MODE_TABLE, uint8_t, atlas.data() )
                else RAYLOD( false, VRC_MODE_TABLE, uint8_t, atlas.data() )
            }
            else if( mode == VRC_MODE_TRILINEAR && voxelBytes == 1 ) CLASSIFY( VRC_MODE_TRILINEAR, uint8_t, atlas.data() )
            else if( mode == VRC_MODE_TRILINEAR ) CLASSIFY( VRC_MODE_TRILINEAR, uint16_t, atlas16.data() )
            else if( mode == VRC_MODE_POINT && voxelBytes == 1 ) CLASSIFY( VRC_MODE_POINT, uint8_t, atlas.data() )
            else if( mode == VRC_MODE_POINT ) CLASSIFY( VRC_MODE_POINT, uint16_t, atlas16.data() )
            else if( dda )
            {
                if( t.clamp ) vrc_pixel_grid_dda< true, true, false, VRC_MODE_TABLE, uint8_t >( ARGS_DDA( atlas.data() ) );
                else if( fixed && g_parts > 1 )
                {
                    /* the march in g_parts slices of the ray (ray compaction, vrc_k_raycast_part): one walk per
                     * slice, the colour carried from one to the next */
                    const vrc_ray r = vrc_setup_ray( f, px, py );
                    const uint32_t pos = py * f.width + px;
                    if( !r.hit )
                    {
                        if( f.clearFirst )
                            pb[pos] = vrc_f4{ 0.f, 0.f, 0.f, 0.f };
                    }
                    else
                    {
                        vrc_f4 color = f.clearFirst ? vrc_f4{ 0.f, 0.f, 0.f, 0.f } : pb[pos];
                        bool alive = !( color.w > VRC_EARLY_EXIT );
                        for( int p = 0; p < g_parts && alive; ++p )
                        {
                            alive = vrc_ray_grid_dda< false, true, true, VRC_MODE_TABLE, uint8_t >(
                                f, r, t.nodes.data(), t.grid.data(), atlas.data(), table, cls, color, n, p, g_parts, vrc_part_dir( f ) );
                            pb[pos] = color;
                        }
                    }
                }
                else if( fixed ) vrc_pixel_grid_dda< false, true, true, VRC_MODE_TABLE, uint8_t >( ARGS_DDA( atlas.data() ) );
                else vrc_pixel_grid_dda< false, true, false, VRC_MODE_TABLE, uint8_t >( ARGS_DDA( atlas.data() ) );
            }
            else
            {
                if( t.clamp ) vrc_pixel_reference_order< true, true, false, VRC_MODE_TABLE, uint8_t >( ARGS_REF( atlas.data() ) );
                else if( fixed ) vrc_pixel_reference_order< false, true, true, VRC_MODE_TABLE, uint8_t >( ARGS_REF( atlas.data() ) );
                else vrc_pixel_reference_order< false, true, false, VRC_MODE_TABLE, uint8_t >( ARGS_REF( atlas.data() ) );
            }
#undef CLASSIFY
#undef RAYLOD
#undef ARGS_DDA
#undef ARGS_REF
            total += n;
        }
    if( samplesOut )
        *samplesOut = total;
    return 0;
}

/* > 1: kernel 4 (grid walk, fixed-point stepping) marches every ray in this many slices (ray compaction) */
extern "C" void harness_set_parts( int parts ) { g_parts = parts; }

extern "C" int harness_render( const uint8_t* atlasRowMajor, const uint32_t atlasDim[3],
                               const uint32_t slotDim[3], float* pixelBuffer, uint32_t W,
                               uint32_t H, const float* planes, uint32_t nPlanes, const float* tf,
                               const vrc_view_data* view, uint32_t nNodes,
                               const vrc_node_data* nodes, const vrc_render_data* render,
                               int fracBits, int kernel, int pixelOffX, int pixelOffY,
                               uint64_t* samplesOut, int* gridOkOut, int voxelBytes, int variant )
{
    return render_impl( atlasRowMajor, atlasDim, slotDim, pixelBuffer, W, H, planes, nPlanes, tf, view,
                        nNodes, nodes, render, fracBits, kernel, pixelOffX, pixelOffY, samplesOut,
                        gridOkOut, voxelBytes, variant, false, 0.f, 0.f );
}

/* per-ray adaptive LOD (vrc_pixel_ray_lod): kernel 1/3 = classified tables with float / fixed-point
 * stepping, 5 = trilinear, 7 = point sampling with per-sample classification */
extern "C" int harness_render_ray_lod( const uint8_t* atlasRowMajor, const uint32_t atlasDim[3],
                                       const uint32_t slotDim[3], float* pixelBuffer, uint32_t W,
                                       uint32_t H, const float* planes, uint32_t nPlanes, const float* tf,
                                       const vrc_view_data* view, uint32_t nNodes,
                                       const vrc_node_data* nodes, const vrc_render_data* render,
                                       int fracBits, int kernel, uint64_t* samplesOut, int* lodOkOut,
                                       int voxelBytes, float screenSpaceError, float worldSpacePerPixel )
{
    return render_impl( atlasRowMajor, atlasDim, slotDim, pixelBuffer, W, H, planes, nPlanes, tf, view,
                        nNodes, nodes, render, fracBits, kernel, 0, 0, samplesOut, lodOkOut, voxelBytes, 0,
                        true, screenSpaceError, worldSpacePerPixel );
}
