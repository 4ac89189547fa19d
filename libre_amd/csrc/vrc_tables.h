/*
 * vrc_tables.h -- host-side derivation of the kernel's node table and brick grid from the
 * reference-shaped NodeData list (cuda/Renderer.cuh:35-41).  Header-only so that the C ABI
 * (vrc_api.hip) and the CPU unit harness (tests/cpu_harness) run the very same code.
 */
#ifndef VRC_TABLES_H
#define VRC_TABLES_H

#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/vrc_hip.h"
#include "vrc_core.h"

struct vrc_atlas_geom
{
    uint32_t atlasDim[3];
    uint32_t slotDim[3];
    uint32_t slots[3];
};

struct vrc_host_tables
{
    std::vector< vrc_dev_node > nodes;
    std::vector< int32_t > grid;
    bool gridOk = false;
    /* every brick covers exactly one grid cell (one LOD level, regular tree): the along-ray order of the
     * grid walk then equals the reference's centre-distance order for every pair of bricks that share a
     * ray; with bricks of different sizes it need not */
    bool oneCellPerBrick = false;
    bool clamp = false;
    vrc_frame g = {}; /* grid and lod* fields only */
    /* per-ray LOD (vrc_build_lod_tables): one cell -> node table per level in `grid` */
    bool lodOk = false;
    uint32_t lodLevels = 0;
    double finestVoxelWorld = 0.0;
};

inline bool vrc_near_int( double v, double tol, long* out )
{
    const double r = std::floor( v + 0.5 );
    *out = (long)r;
    return std::fabs( v - r ) <= tol;
}

inline void vrc_build_tables( const vrc_atlas_geom& p, const vrc_node_data* in, uint32_t n,
                              vrc_host_tables& t )
{
    t.nodes.resize( n );
    t.clamp = false;
    for( uint32_t i = 0; i < n; ++i )
    {
        const vrc_node_data& s = in[i];
        vrc_dev_node& d = t.nodes[i];
        uint32_t slotIdx[3];
        for( int a = 0; a < 3; ++a )
        {
            d.aabbMin[a] = s.aabbMin[a];
            d.aabbSize[a] = s.aabbSize[a];
            const double dim = (double)p.atlasDim[a];
            double texMinVox = (double)s.textureMin[a] * dim;
            double texSizeVox = (double)s.textureSize[a] * dim;
            /* textureMin/Size are float32 fractions of the atlas (CudaTextureObject.cpp:77-82),
             * so times the atlas size they miss the integer voxel they stand for by ~1e-4, by an
             * amount that depends on which slot the brick happened to get.  Snap them back:
             * frames then do not depend on the (thread-timing dependent) slot assignment. */
            if( std::fabs( texMinVox - std::floor( texMinVox + 0.5 ) ) < 1e-2 )
                texMinVox = std::floor( texMinVox + 0.5 );
            if( std::fabs( texSizeVox - std::floor( texSizeVox + 0.5 ) ) < 1e-2 )
                texSizeVox = std::floor( texSizeVox + 0.5 );
            d.voxPerWorld[a] = (float)( texSizeVox / (double)s.aabbSize[a] );
            long tv = (long)std::floor( texMinVox + 0.5 );
            if( tv < 0 ) tv = 0;
            if( tv > (long)p.atlasDim[a] - 1 ) tv = (long)p.atlasDim[a] - 1;
            slotIdx[a] = (uint32_t)tv / p.slotDim[a];
            d.localOrigin[a] = (float)( texMinVox - (double)( slotIdx[a] * p.slotDim[a] ) );
            /* samples may land one voxel outside the interior on either side; if that can
             * leave the slot (overlap 0), the kernel clamps inside the slot */
            if( d.localOrigin[a] < 1.0f ||
                d.localOrigin[a] + (float)texSizeVox > (float)p.slotDim[a] - 1.0f )
                t.clamp = true;
        }
        vrc_layout lay;
        for( int a = 0; a < 3; ++a )
        {
            lay.slots[a] = p.slots[a];
            lay.slotDim[a] = p.slotDim[a];
        }
        const uint64_t base = vrc_slot_base( lay, slotIdx[0], slotIdx[1], slotIdx[2] );
        d.slotBase = (uint32_t)base;
        d.slotBaseHi = (uint32_t)( base >> 32 );
        d.level = 0;
        d.pad = 0;
    }

    /* brick grid: cells of the finest brick size covering the union of the node boxes */
    t.gridOk = false;
    t.grid.clear();
    for( int a = 0; a < 3; ++a )
    {
        t.g.gridMin[a] = 0.f;
        t.g.cellSize[a] = 1.f;
        t.g.invCellSize[a] = 1.f;
        t.g.gridDim[a] = 0;
    }
    if( n == 0 )
        return;
    double cell[3], gmin[3], gmax[3];
    for( int a = 0; a < 3; ++a )
    {
        cell[a] = in[0].aabbSize[a];
        gmin[a] = in[0].aabbMin[a];
        gmax[a] = (double)in[0].aabbMin[a] + in[0].aabbSize[a];
    }
    for( uint32_t i = 1; i < n; ++i )
        for( int a = 0; a < 3; ++a )
        {
            cell[a] = std::min( cell[a], (double)in[i].aabbSize[a] );
            gmin[a] = std::min( gmin[a], (double)in[i].aabbMin[a] );
            gmax[a] = std::max( gmax[a], (double)in[i].aabbMin[a] + in[i].aabbSize[a] );
        }
    long dim[3];
    for( int a = 0; a < 3; ++a )
    {
        if( !( cell[a] > 0.0 ) )
            return;
        if( !vrc_near_int( ( gmax[a] - gmin[a] ) / cell[a], 1e-3, &dim[a] ) || dim[a] < 1 ||
            dim[a] > 4096 )
            return;
    }
    if( (double)dim[0] * dim[1] * dim[2] > 64.0 * 1024 * 1024 )
        return;
    t.grid.assign( (size_t)dim[0] * dim[1] * dim[2], -1 );
    bool oneCell = true;
    for( uint32_t i = 0; i < n; ++i )
    {
        long i0[3], cnt[3];
        for( int a = 0; a < 3; ++a )
        {
            if( !vrc_near_int( ( (double)in[i].aabbMin[a] - gmin[a] ) / cell[a], 1e-3, &i0[a] ) ||
                !vrc_near_int( (double)in[i].aabbSize[a] / cell[a], 1e-3, &cnt[a] ) || cnt[a] < 1 ||
                i0[a] < 0 || i0[a] + cnt[a] > dim[a] )
            {
                t.grid.clear();
                return;
            }
        }
        oneCell = oneCell && cnt[0] == 1 && cnt[1] == 1 && cnt[2] == 1;
        for( long z = i0[2]; z < i0[2] + cnt[2]; ++z )
            for( long y = i0[1]; y < i0[1] + cnt[1]; ++y )
                for( long x = i0[0]; x < i0[0] + cnt[0]; ++x )
                {
                    int32_t& c = t.grid[( (size_t)z * dim[1] + y ) * dim[0] + x];
                    if( c != -1 )
                    {
                        /* overlapping nodes: not a partition, use the generic kernel */
                        t.grid.clear();
                        return;
                    }
                    c = (int32_t)i;
                }
    }
    for( int a = 0; a < 3; ++a )
    {
        t.g.gridMin[a] = (float)gmin[a];
        t.g.cellSize[a] = (float)cell[a];
        t.g.invCellSize[a] = (float)( 1.0 / cell[a] );
        t.g.gridDim[a] = (int32_t)dim[a];
    }
    t.gridOk = true;
    t.oneCellPerBrick = oneCell;
}

/* Per-ray LOD tables (vrc_pixel_ray_lod): the node list is a hierarchy -- boxes of different
 * levels may nest, a level is a regular grid of bricks anchored at the min corner of all boxes,
 * with at most one brick per cell (border bricks may be smaller than the cell; the levels of a
 * ragged tree such as UVF's need not align with each other).  Level of a node = rank of its voxel
 * size among the sizes in the list (sizes within 5 % are one level), robust for ragged border
 * bricks whose box size says nothing about their level.  t.grid holds the cell -> node table of
 * every level, t.g the grids (lod* fields; gridMin / gridDim / cellSize = level 0).  Overwrites
 * t.grid / t.g; t.lodOk = false (and t.grid empty) when the list does not fit this form. */
inline void vrc_build_lod_tables( const vrc_atlas_geom& p, const vrc_node_data* in, uint32_t n,
                                  vrc_host_tables& t )
{
    t.lodOk = false;
    t.lodLevels = 0;
    t.grid.clear();
    if( n == 0 || t.nodes.size() != n )
        return;
    std::vector< double > vw( n );
    double vw0 = 0.0, gmin[3], gmax[3];
    for( uint32_t i = 0; i < n; ++i )
    {
        const double texVox = std::floor( (double)in[i].textureSize[0] * p.atlasDim[0] + 0.5 );
        if( !( texVox >= 1.0 ) || !( in[i].aabbSize[0] > 0.f ) )
            return;
        vw[i] = (double)in[i].aabbSize[0] / texVox;
        vw0 = i == 0 ? vw[i] : std::min( vw0, vw[i] );
        for( int a = 0; a < 3; ++a )
        {
            const double lo = in[i].aabbMin[a], hi = lo + (double)in[i].aabbSize[a];
            gmin[a] = i == 0 ? lo : std::min( gmin[a], lo );
            gmax[a] = i == 0 ? hi : std::max( gmax[a], hi );
        }
    }
    /* levels: clusters of voxel sizes, finest first */
    double rep[VRC_MAX_LOD_LEVELS];
    uint32_t levels = 0;
    for( ;; )
    {
        double next = 0.0;
        for( uint32_t i = 0; i < n; ++i )
            if( ( levels == 0 || vw[i] > rep[levels - 1] * 1.05 ) && ( next == 0.0 || vw[i] < next ) )
                next = vw[i];
        if( next == 0.0 )
            break;
        if( levels == VRC_MAX_LOD_LEVELS )
            return;
        rep[levels++] = next;
    }
    double cell[VRC_MAX_LOD_LEVELS][3] = {};
    for( uint32_t i = 0; i < n; ++i )
    {
        uint32_t lv = 0;
        while( lv + 1 < levels && vw[i] > rep[lv] * 1.05 )
            ++lv;
        t.nodes[i].level = lv;
        for( int a = 0; a < 3; ++a )
            cell[lv][a] = std::max( cell[lv][a], (double)in[i].aabbSize[a] );
    }
    size_t total = 0;
    for( uint32_t lv = 0; lv < levels; ++lv )
    {
        for( int a = 0; a < 3; ++a )
        {
            if( !( cell[lv][a] > 0.0 ) )
                return;
            const double cnt = std::ceil( ( gmax[a] - gmin[a] ) / cell[lv][a] - 1e-3 );
            if( cnt < 1.0 || cnt > 4096.0 )
                return;
            t.g.lodDim[lv][a] = (int32_t)cnt;
            t.g.lodInvCell[lv][a] = (float)( 1.0 / cell[lv][a] );
        }
        t.g.lodTable[lv] = (uint32_t)total;
        total += (size_t)t.g.lodDim[lv][0] * t.g.lodDim[lv][1] * t.g.lodDim[lv][2];
        if( total > 64u * 1024u * 1024u )
            return;
    }
    t.grid.assign( total, -1 );
    for( uint32_t i = 0; i < n; ++i )
    {
        const uint32_t lv = t.nodes[i].level;
        long idx[3];
        for( int a = 0; a < 3; ++a )
            if( !vrc_near_int( ( (double)in[i].aabbMin[a] - gmin[a] ) / cell[lv][a], 1e-3, &idx[a] ) ||
                idx[a] < 0 || idx[a] >= t.g.lodDim[lv][a] )
            {
                t.grid.clear();
                return;
            }
        int32_t& c = t.grid[t.g.lodTable[lv] +
                            ( (size_t)idx[2] * t.g.lodDim[lv][1] + idx[1] ) * t.g.lodDim[lv][0] + idx[0]];
        if( c != -1 ) /* two bricks of one level in one cell */
        {
            t.grid.clear();
            return;
        }
        c = (int32_t)i;
    }
    for( int a = 0; a < 3; ++a )
    {
        t.g.gridMin[a] = (float)gmin[a];
        t.g.lodMax[a] = (float)gmax[a];
        t.g.cellSize[a] = (float)cell[0][a];
        t.g.invCellSize[a] = t.g.lodInvCell[0][a];
        t.g.gridDim[a] = t.g.lodDim[0][a];
    }
    t.g.lodEps = (float)( vw0 * 0.01 );
    t.lodLevels = levels;
    t.finestVoxelWorld = vw0;
    t.lodOk = true;
}

/* Frame constants from the reference-shaped PODs (Renderer.cu:159-170 derives the same
 * values per thread).  Grid fields come from vrc_build_tables. */
inline void vrc_fill_frame( vrc_frame& f, const vrc_view_data& view, const vrc_render_data& render,
                            const vrc_atlas_geom& geom, const vrc_frame& gridFrame,
                            const float planes[6][4], uint32_t nPlanes, uint32_t nNodes,
                            uint32_t fbW, uint32_t fbH, float pixelOffX, float pixelOffY )
{
    for( int i = 0; i < 3; ++i )
    {
        f.eye[i] = view.eyePosition[i];
        f.aabbMin[i] = view.aabbMin[i];
        f.aabbMax[i] = view.aabbMax[i];
        f.slotDim[i] = geom.slotDim[i];
        f.gridMin[i] = gridFrame.gridMin[i];
        f.cellSize[i] = gridFrame.cellSize[i];
        f.invCellSize[i] = gridFrame.invCellSize[i];
        f.gridDim[i] = gridFrame.gridDim[i];
        f.lodMax[i] = gridFrame.lodMax[i];
    }
    f.lodEps = gridFrame.lodEps;
    for( int lv = 0; lv < VRC_MAX_LOD_LEVELS; ++lv )
    {
        f.lodTable[lv] = gridFrame.lodTable[lv];
        for( int i = 0; i < 3; ++i )
        {
            f.lodInvCell[lv][i] = gridFrame.lodInvCell[lv][i];
            f.lodDim[lv][i] = gridFrame.lodDim[lv][i];
        }
    }
    f.vpX = (float)view.glViewport[0];
    f.vpY = (float)view.glViewport[1];
    f.vpW = (float)view.glViewport[2];
    f.vpH = (float)view.glViewport[3];
    f.pixelOffX = pixelOffX;
    f.pixelOffY = pixelOffY;
    for( int i = 0; i < 16; ++i )
    {
        f.invProj[i] = view.invProjMatrix[i];
        f.invView[i] = view.invViewMatrix[i];
    }
    f.nearPlane = view.nearPlane;
    /* Renderer.cu:170: 1.0 / float(spr) evaluated in double, stored to float */
    f.stepSize = (float)( 1.0 / (double)(float)render.samplesPerRay );
    f.width = fbW;
    f.height = fbH;
    f.nPlanes = nPlanes;
    for( int i = 0; i < 6; ++i )
        for( int k = 0; k < 4; ++k )
            f.planes[i][k] = planes[i][k];
    f.nodeCount = nNodes;
    f.sbx = geom.slotDim[0] / VRC_MB;
    f.sby = geom.slotDim[1] / VRC_MB;
    f.rowMap = nullptr;
    f.samplesPerPixel = 1u; /* the caller sets it for the glRaycaster variant */
    f.lodLevels = 0;
    f.lodBase = 0.f;
}

#endif
