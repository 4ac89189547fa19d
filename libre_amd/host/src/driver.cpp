/* driver.cpp -- implementation of include/livre_hip_driver.h (headless apps/livre + livre/eq
 * Channel::frameDraw around the plugin surface). */
#include "livre_hip_driver.h"

#include <cstring>

#include "livre_hip/hip.h"

using namespace livre;

namespace
{
thread_local std::string g_error;
int fail( const std::string& m )
{
    g_error = m;
    return 1;
}
const float nearPlane = 0.1f; /* livre/eq/Channel.cpp:61-62 */
const float farPlane = 15.0f;
}

struct lvh_app
{
    std::unique_ptr< DataSource > dataSource;
    std::unique_ptr< RenderPipeline > pipeline;
    lvh_params params;
    CameraSettings camera;
    RenderSettings renderSettings;
    RendererParameters vrParameters;
    uint32_t frameId = 0;
    uint32_t timeStep = 0;                 /* lvh_app_set_time_step: FrameInfo::timeStep (livre/eq/Channel.cpp:259-270) */
    RenderStatistics lastStats;
    std::vector< uint32_t > rowMap; /* lvh_app_set_bands */
    std::vector< std::unique_ptr< Renderer > > extraRenderers; /* frames in flight beyond the first */
    uint32_t slot = 0;
    std::string rendererName;
    float dataRange[2] = { 0.0f, 0.0f }; /* lvh_app_set_data_range; empty = the voxel type's range */
    vrc_comm* comm = nullptr;              /* lvh_app_comm_create: sort-first tile exchange */
    std::vector< vrc_band > layout;        /* lvh_app_set_layout: every band of the frame, all ranks */

    Renderer& currentRenderer()
    {
        return slot == 0 ? pipeline->getRenderer() : *extraRenderers[slot - 1];
    }
    HipRaycastRenderer& renderer()
    {
        return static_cast< HipRaycastRenderer& >( currentRenderer().getPlugin() );
    }
    HipRaycastPipeline& hipPipeline() { return static_cast< HipRaycastPipeline& >( pipeline->getPlugin() ); }

    void tile( uint32_t t[4] ) const
    {
        if( !rowMap.empty() || params.tile[2] == 0 || params.tile[3] == 0 )
        {
            t[0] = t[1] = 0;
            t[2] = params.width;
            t[3] = params.height;
        }
        else
            std::memcpy( t, params.tile, sizeof( params.tile ) );
    }

    /* Channel::setupFrustum (livre/eq/Channel.cpp:151-164): the channel's sub-frustum of the
     * wall frustum l/r/b/t = -/+0.05 at near 0.1 */
    Matrix4f projection() const
    {
        uint32_t t[4];
        tile( t );
        const float l = -0.05f, r = 0.05f, b = -0.05f, tp = 0.05f;
        const float W = float( params.width ), H = float( params.height );
        const float tl = l + ( r - l ) * float( t[0] ) / W;
        const float tr = l + ( r - l ) * float( t[0] + t[2] ) / W;
        const float tb = b + ( tp - b ) * float( t[1] ) / H;
        const float tt = b + ( tp - b ) * float( t[1] + t[3] ) / H;
        return perspectiveFrustum( tl, tr, tb, tt, nearPlane, farPlane );
    }

    /** (0,255) is hard-coded at Channel.cpp:284; 16-bit volumes (extension) get the type's
     *  range unless lvh_app_set_data_range gave one */
    Vector2f dataSourceRange() const
    {
        if( dataRange[1] > dataRange[0] )
            return Vector2f( dataRange[0], dataRange[1] );
        const size_t bytes = dataSource->getVolumeInfo().getBytesPerVoxel();
        return bytes == 2 ? Vector2f( 0.0f, 65535.0f ) : Vector2f( 0.0f, 255.0f );
    }

    RenderInputs inputs()
    {
        uint32_t t[4];
        tile( t );
        const Frustum frustum( camera.getModelViewMatrix(), projection() );
        return RenderInputs{ FrameInfo( frustum, timeStep, frameId ),
                             Range{ { 0.0f, 1.0f } },
                             dataSourceRange(),
                             PixelViewport( 0, 0, int32_t( t[2] ), int32_t( t[3] ) ),
                             Viewport( float( t[0] ) / params.width, float( t[1] ) / params.height,
                                       float( t[2] ) / params.width, float( t[3] ) / params.height ),
                             renderSettings,
                             vrParameters,
                             nullptr,
                             *dataSource,
                             rowMap };
    }
};

extern "C" {

const char* lvh_last_error( void ) { return g_error.c_str(); }

int lvh_app_create( const char* uri, const char* rendererName, const lvh_params* p, lvh_app** out )
{
    if( !uri || !rendererName || !p || !out )
        return fail( "lvh_app_create: NULL argument" );
    *out = nullptr;
    try
    {
        std::unique_ptr< lvh_app > app( new lvh_app() );
        app->params = *p;
        if( p->width == 0 || p->height == 0 )
            return fail( "lvh_app_create: empty frame" );
        setHipDevice( p->device );
        app->dataSource.reset( new DataSource( std::string( uri ) ) );
        app->pipeline.reset( new RenderPipeline( rendererName ) );
        app->rendererName = rendererName;
        RendererParameters& v = app->vrParameters;
        v.synchronousMode = p->synchronous != 0;
        v.samplesPerRay = p->samples_per_ray;
        v.minLOD = p->min_lod;
        if( p->max_lod ) v.maxLOD = p->max_lod;
        if( p->sse > 0.f ) v.screenSpaceError = p->sse;
        if( p->gpu_cache_mb ) v.maxGPUCacheMemoryMB = p->gpu_cache_mb;
        if( p->cpu_cache_mb ) v.maxCPUCacheMemoryMB = p->cpu_cache_mb;
        /* ApplicationParameters.cpp:54-55 + Config: position then look-at */
        app->camera.setCameraPosition( Vector3f( 0.f, 0.f, 1.5f ) );
        app->camera.setCameraLookAt( Vector3f( 0.f, 0.f, 0.f ) );
        *out = app.release();
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

void lvh_app_destroy( lvh_app* app )
{
    if( app && app->comm )
        vrc_comm_destroy( app->comm );
    delete app;
}

int lvh_app_set_camera( lvh_app* app, const float pos[3], const float lookat[3], float sx, float sy )
{
    if( !app || !pos || !lookat ) return fail( "NULL argument" );
    app->camera = CameraSettings();
    app->camera.setCameraPosition( Vector3f( pos[0], pos[1], pos[2] ) );
    app->camera.setCameraLookAt( Vector3f( lookat[0], lookat[1], lookat[2] ) );
    app->camera.spinModel( sx, sy );
    return 0;
}

int lvh_app_set_time_step( lvh_app* app, uint32_t timeStep )
{
    if( !app ) return fail( "NULL argument" );
    const Vector2ui range = app->dataSource->getVolumeInfo().frameRange;
    if( timeStep < range[0] || timeStep >= range[1] )
        return fail( "lvh_app_set_time_step: time step outside the data source's frame range" );
    app->timeStep = timeStep;
    return 0;
}

int lvh_app_set_modelview( lvh_app* app, const float mv[16] )
{
    if( !app || !mv ) return fail( "NULL argument" );
    app->camera.setModelViewMatrix( Matrix4f( mv, mv + 16 ) );
    return 0;
}

int lvh_app_set_colormap( lvh_app* app, const float rgba[1024] )
{
    if( !app || !rgba ) return fail( "NULL argument" );
    app->renderSettings.getColorMap().setSamples( rgba );
    return 0;
}

int lvh_app_set_clip_planes( lvh_app* app, const float* planes, uint32_t n )
{
    if( !app || ( n && !planes ) ) return fail( "NULL argument" );
    app->renderSettings.getClipPlanes().clear();
    for( uint32_t i = 0; i < n; ++i )
        app->renderSettings.getClipPlanes().addPlane(
            Vector4f( planes[i * 4], planes[i * 4 + 1], planes[i * 4 + 2], planes[i * 4 + 3] ) );
    return 0;
}

#define LVH_TRY( stmt )                   \
    try                                   \
    {                                     \
        stmt;                             \
        return 0;                         \
    }                                     \
    catch( const std::exception& e )      \
    {                                     \
        return fail( e.what() );          \
    }

int lvh_app_set_frames_in_flight( lvh_app* app, uint32_t n )
{
    if( !app || n == 0 || n > 8 ) return fail( "lvh_app_set_frames_in_flight: 1..8 expected" );
    try
    {
        while( app->extraRenderers.size() + 1 < n )
            app->extraRenderers.emplace_back( new Renderer( app->rendererName ) );
        while( app->extraRenderers.size() + 1 > n )
            app->extraRenderers.pop_back();
        if( app->slot >= n )
            app->slot = 0;
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

int lvh_app_select_slot( lvh_app* app, uint32_t slot )
{
    if( !app || slot > app->extraRenderers.size() ) return fail( "lvh_app_select_slot: no such slot" );
    app->slot = slot;
    return 0;
}

int lvh_app_set_option( lvh_app* app, int option, int64_t value )
{
    if( !app ) return fail( "NULL argument" );
    LVH_TRY( app->renderer().setOption( option, value ) )
}
int lvh_app_set_data_range( lvh_app* app, float lo, float hi )
{
    if( !app ) return fail( "NULL argument" );
    if( !( hi > lo ) ) return fail( "empty data range" );
    app->dataRange[0] = lo;
    app->dataRange[1] = hi;
    return 0;
}
int lvh_app_set_ray_lod( lvh_app* app, int enable )
{
    if( !app ) return fail( "NULL argument" );
    app->vrParameters.rayLOD = enable != 0;
    return 0;
}
int lvh_app_set_stream( lvh_app* app, void* s )
{
    if( !app ) return fail( "NULL argument" );
    LVH_TRY( app->renderer().setStream( s ) )
}
int lvh_app_set_framebuffer( lvh_app* app, void* d )
{
    if( !app ) return fail( "NULL argument" );
    uint32_t t[4];
    app->tile( t );
    const uint32_t rows = app->rowMap.empty() ? t[3] : uint32_t( app->rowMap.size() );
    LVH_TRY( app->renderer().setFrameBuffer( d, t[2], rows ) )
}

int lvh_app_set_bands( lvh_app* app, const uint32_t* y0, const uint32_t* h, uint32_t n )
{
    if( !app || ( n && ( !y0 || !h ) ) ) return fail( "NULL argument" );
    std::vector< uint32_t > rows;
    for( uint32_t i = 0; i < n; ++i )
        for( uint32_t r = 0; r < h[i]; ++r )
        {
            if( y0[i] + r >= app->params.height )
                return fail( "lvh_app_set_bands: band outside the frame" );
            rows.push_back( y0[i] + r );
        }
    app->rowMap.swap( rows );
    return 0;
}

static void fillStats( lvh_app* app, lvh_frame_stats* s, bool sync )
{
    std::memset( s, 0, sizeof( *s ) );
    s->n_available = app->lastStats.nAvailable;
    s->n_not_available = app->lastStats.nNotAvailable;
    s->n_render_available = app->lastStats.nRenderAvailable;
    s->n_passes = app->hipPipeline().lastNumberOfPasses();
    s->ray_lod = ( app->hipPipeline().lastFrameUsedRayLOD() && app->renderer().lastRenderUsedRayLOD() ) ? 1u : 0u;
    s->samples_per_ray = app->renderer().getComputedSamplesPerRay();
    if( sync )
        app->renderer().kernelStats( &s->kernel_ms, &s->kernel_ms_sum, &s->kernel_launches, &s->samples );
}

int lvh_app_render_frame( lvh_app* app, float* host, lvh_frame_stats* stats )
{
    if( !app ) return fail( "NULL argument" );
    try
    {
        const RenderInputs in = app->inputs();
        app->lastStats = app->pipeline->getPlugin().render( app->currentRenderer(), in );
        ++app->frameId;
        if( host )
            app->renderer().readFrame( host );
        if( stats )
            fillStats( app, stats, false );
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

int lvh_app_get_stats( lvh_app* app, lvh_frame_stats* stats )
{
    if( !app || !stats ) return fail( "NULL argument" );
    LVH_TRY( fillStats( app, stats, true ) )
}
int lvh_app_wait_uploads( lvh_app* app )
{
    if( !app ) return fail( "NULL argument" );
    LVH_TRY( app->hipPipeline().waitForUploads() )
}
int lvh_app_synchronize( lvh_app* app )
{
    if( !app ) return fail( "NULL argument" );
    LVH_TRY( app->renderer().synchronize() )
}

int lvh_app_volume_info( lvh_app* app, uint32_t voxels[3], uint32_t maxBlock[3], uint32_t overlap[3],
                         float worldSize[3], uint32_t* depth, uint32_t rootBlocks[3] )
{
    if( !app ) return fail( "NULL argument" );
    const VolumeInformation& v = app->dataSource->getVolumeInfo();
    for( int i = 0; i < 3; ++i )
    {
        if( voxels ) voxels[i] = v.voxels[i];
        if( maxBlock ) maxBlock[i] = v.maximumBlockSize[i];
        if( overlap ) overlap[i] = v.overlap[i];
        if( worldSize ) worldSize[i] = v.worldSize[i];
        if( rootBlocks ) rootBlocks[i] = v.rootNode.getBlockSize()[i];
    }
    if( depth ) *depth = v.rootNode.getDepth();
    return 0;
}

static int copyIds( const NodeIds& ids, uint64_t* out, size_t cap, size_t* n )
{
    if( n ) *n = ids.size();
    if( out )
        for( size_t i = 0; i < ids.size() && i < cap; ++i )
            out[i] = ids[i].getId();
    return 0;
}

int lvh_app_visible_set( lvh_app* app, uint64_t* ids, size_t cap, size_t* n )
{
    if( !app ) return fail( "NULL argument" );
    try
    {
        const RenderInputs in = app->inputs();
        const RendererParameters& p = in.vrParameters;
        SelectVisibles visitor( in.dataSource, in.frameInfo.frustum, uint32_t( in.pixelViewPort[3] ),
                                p.getSSE(), p.getMinLOD(), p.getMaxLOD(), in.renderDataRange,
                                in.renderSettings.getClipPlanes() );
        DFSTraversal traverser;
        traverser.traverse( in.dataSource.getVolumeInfo().rootNode, visitor, 0 );
        return copyIds( visitor.getVisibles(), ids, cap, n );
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

/* ---- sort-first tile exchange: the C ABI's RCCL gather driven from the host side ------------------- */
int lvh_comm_unique_id( uint8_t id[128] )
{
    if( vrc_comm_unique_id( id ) != VRC_OK ) return fail( vrc_last_error() );
    return 0;
}

int lvh_app_comm_create( lvh_app* app, int rank, int world, const uint8_t* id )
{
    if( !app ) return fail( "NULL argument" );
    if( app->comm )
    {
        vrc_comm_destroy( app->comm );
        app->comm = nullptr;
    }
    if( vrc_comm_create( app->renderer().deviceContext(), rank, world, id, &app->comm ) != VRC_OK )
        return fail( vrc_last_error() );
    return 0;
}

int lvh_app_set_layout( lvh_app* app, const uint32_t* rank, const uint32_t* y0, const uint32_t* h, uint32_t n )
{
    if( !app || ( n && ( !rank || !y0 || !h ) ) ) return fail( "NULL argument" );
    std::vector< vrc_band > l( n );
    for( uint32_t i = 0; i < n; ++i )
    {
        if( uint64_t( y0[i] ) + uint64_t( h[i] ) > uint64_t( app->params.height ) ) /* 64 bits: y0 = 0xFFFFFFFF, h = 2 */
            return fail( "lvh_app_set_layout: band outside the frame" );
        l[i].rank = rank[i];
        l[i].frame_row = y0[i];
        l[i].rows = h[i];
    }
    app->layout.swap( l );
    return 0;
}

int lvh_app_gather_tiles( lvh_app* app, uint32_t nFrames, const void* localDevice, size_t localFrameStride,
                          void* frameDevice, size_t frameStride, int root, void* hipStream )
{
    if( !app || !app->comm ) return fail( "lvh_app_gather_tiles: no communicator (lvh_app_comm_create)" );
    if( vrc_gather_tiles( app->renderer().deviceContext(), app->comm, app->layout.data(),
                          uint32_t( app->layout.size() ), app->params.width, app->params.height, nFrames, localDevice,
                          localFrameStride, frameDevice, frameStride, root, hipStream ) != VRC_OK )
        return fail( vrc_last_error() );
    return 0;
}

int lvh_app_node_order( lvh_app* app, uint64_t* ids, size_t cap, size_t* n )
{
    if( !app ) return fail( "NULL argument" );
    const std::vector< Identifier >& order = app->renderer().lastNodeOrder();
    if( n ) *n = order.size();
    if( ids )
        for( size_t i = 0; i < order.size() && i < cap; ++i )
            ids[i] = order[i];
    return 0;
}

int lvh_app_view_matrices( lvh_app* app, float mv[16], float proj[16] )
{
    if( !app ) return fail( "NULL argument" );
    if( mv ) std::memcpy( mv, app->camera.getModelViewMatrix().array, 64 );
    if( proj ) std::memcpy( proj, app->projection().array, 64 );
    return 0;
}

int lvh_app_cache_stats( lvh_app* app, uint64_t tex[4], uint64_t data[4] )
{
    if( !app ) return fail( "NULL argument" );
    const CacheStatistics* t = app->hipPipeline().textureCacheStatistics();
    const CacheStatistics* d = app->hipPipeline().dataCacheStatistics();
    if( tex && t ) { tex[0] = t->getUsedMemory(); tex[1] = t->getMaximumMemory(); tex[2] = t->getBlockCount(); tex[3] = t->getMisses(); }
    if( data && d ) { data[0] = d->getUsedMemory(); data[1] = d->getMaximumMemory(); data[2] = d->getBlockCount(); data[3] = d->getMisses(); }
    return ( t && d ) ? 0 : fail( "caches not created yet" );
}

int lvh_select_visibles( const char* uri, const float mv[16], const float proj[16], uint32_t windowHeight,
                         float sse, uint32_t minLOD, uint32_t maxLOD, uint64_t* ids, size_t cap, size_t* n )
{
    try
    {
        DataSource dataSource{ std::string( uri ) };
        const Frustum frustum( Matrix4f( mv, mv + 16 ), Matrix4f( proj, proj + 16 ) );
        ClipPlanes planes; /* tests/lib/lodSelection.cpp:52: default-constructed = six planes */
        SelectVisibles visitor( dataSource, frustum, windowHeight, sse, minLOD, maxLOD,
                                Range{ { 0.0f, 1.0f } }, planes );
        DFSTraversal traverser;
        traverser.traverse( dataSource.getVolumeInfo().rootNode, visitor, 0 );
        return copyIds( visitor.getVisibles(), ids, cap, n );
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

/* tests/core/cache.cpp:33-75 restated against the mirrored Cache<T> */
namespace
{
class ValidCacheObject : public CacheObject
{
public:
    explicit ValidCacheObject( const CacheId& id ) : CacheObject( id ) {}
    size_t getSize() const final { return 1000; }
};
}

int lvh_selftest_cache( void )
{
#define CHECK( n, cond ) if( !( cond ) ) { fail( "cache selftest: " #cond ); return n; }
    Cache< ValidCacheObject > cache( "Test Cache", 2048u );
    CHECK( 1, cache.getCount() == 0 );
    CHECK( 2, cache.getStatistics().getMaximumMemory() == 2048u );
    CHECK( 3, !cache.get( 1 ) );
    ConstCacheObjectPtr obj = cache.load( 1 );
    size_t size = obj->getSize();
    CHECK( 4, cache.getStatistics().getUsedMemory() == size );
    obj = cache.load( 2 );
    CHECK( 5, obj && cache.getCount() == 2 && obj->getId() == 2 && obj.use_count() == 2 );
    size += obj->getSize();
    CHECK( 6, cache.getStatistics().getUsedMemory() == size );
    obj = cache.load( 1 );
    CHECK( 7, obj && cache.getCount() == 2 && obj->getId() == 1 && obj.use_count() == 2 );
    CHECK( 8, cache.getStatistics().getUsedMemory() == size );
    obj.reset();
    ConstCacheObjectPtr trigger = cache.load( 3 );
    CHECK( 9, trigger && cache.getCount() == 2 && trigger->getId() == 3 && trigger.use_count() == 2 );
    CHECK( 10, cache.getStatistics().getUsedMemory() == size );
    /* a referenced object is never evicted */
    ConstCacheObjectPtr held = cache.load( 4 );
    ConstCacheObjectPtr held2 = cache.load( 5 );
    CHECK( 11, cache.get( 3 ) && cache.get( 4 ) && cache.get( 5 ) ); /* all referenced: over budget, kept */
    cache.purge();
    CHECK( 12, cache.getCount() == 0 && cache.getStatistics().getUsedMemory() == 0 );
    CHECK( 13, !cache.load( INVALID_CACHE_ID ) );
    return 0;
}

/* tests/core/pluginFactory.cpp:115-203: no plugin -> runtime_error; first handles()==true wins */
namespace
{
struct TestPlugin
{
    typedef TestPlugin PluginT;
    explicit TestPlugin( const int& v ) : value( v ) {}
    virtual ~TestPlugin() {}
    virtual int id() const = 0;
    int value;
};
struct PluginA : TestPlugin
{
    explicit PluginA( const int& v ) : TestPlugin( v ) {}
    static bool handles( const int& v ) { return v < 10; }
    int id() const final { return 1; }
};
struct PluginB : TestPlugin
{
    explicit PluginB( const int& v ) : TestPlugin( v ) {}
    static bool handles( const int& v ) { return v < 100; }
    int id() const final { return 2; }
};
}

int lvh_selftest_plugin_factory( void )
{
    typedef PluginFactory< TestPlugin, const int& > Factory;
    Factory& f = Factory::getInstance();
    f.deregisterAll();
    try { f.create( 1 ); fail( "factory created without plugins" ); return 1; }
    catch( const std::runtime_error& ) {}
    {
        PluginRegisterer< PluginA, const int& > a;
        PluginRegisterer< PluginB, const int& > b;
        std::unique_ptr< TestPlugin > p( f.create( 5 ) );
        CHECK( 2, p->id() == 1 && p->value == 5 );
        p.reset( f.create( 50 ) );
        CHECK( 3, p->id() == 2 );
        try { f.create( 500 ); fail( "factory created for unhandled data" ); return 4; }
        catch( const std::runtime_error& ) {}
    }
    f.deregisterAll();
    /* the renderer registry: unknown name throws, "hip" is registered by this library */
    try { Renderer r( "no-such-renderer" ); fail( "unknown renderer accepted" ); return 5; }
    catch( const std::runtime_error& ) {}
    CHECK( 6, HipRaycastRenderer::handles( "hip" ) && !HipRaycastRenderer::handles( "cuda" ) );
    return 0;
#undef CHECK
}

/* tests/eq/settings/cameraSettings.cpp:42-144 cases: spin, lookat, everything, default app view */
int lvh_selftest_camera( float out[4][16] )
{
    CameraSettings a;
    a.spinModel( 20.0f, 20.0f );
    std::memcpy( out[0], a.getModelViewMatrix().array, 64 );
    CameraSettings b;
    b.setCameraLookAt( Vector3f( 20.0f, 20.0f, 20.0f ) );
    std::memcpy( out[1], b.getModelViewMatrix().array, 64 );
    CameraSettings c;
    c.setCameraPosition( Vector3f( 0.5f, 1.17f, 6.78f ) );
    c.setCameraLookAt( Vector3f( 13.52f, 123.53f, 21.12f ) );
    c.spinModel( 13.54f, 21.49f );
    c.moveCamera( 13.54f, 21.49f, 33.25f );
    std::memcpy( out[2], c.getModelViewMatrix().array, 64 );
    CameraSettings d;
    d.setCameraPosition( Vector3f( 0.f, 0.f, 1.5f ) );
    d.setCameraLookAt( Vector3f( 0.f, 0.f, 0.f ) );
    std::memcpy( out[3], d.getModelViewMatrix().array, 64 );
    return 0;
}

/* tests/core/clipPlanes.cpp:29-59 (testClipping), statement by statement; returns the line of the first failed check */
int lvh_selftest_clip_planes( void )
{
#define LVH_CHECK( cond ) if( !( cond ) ) return __LINE__
    const Boxf boxInside( Vector3f( -0.3f, -0.3f, -0.3f ), Vector3f( 0.3f, 0.3f, 0.3f ) );
    const Boxf boxOutside( Vector3f( 0.8f, 0.8f, 0.8f ), Vector3f( 0.9f, 0.9f, 0.9f ) );
    const Boxf boxIntersect( Vector3f( -0.3f, -0.3f, -0.3f ), Vector3f( 0.9f, 0.9f, 0.9f ) );
    ClipPlanes clipPlanes;
    LVH_CHECK( !clipPlanes.isClipped( boxInside ) );
    LVH_CHECK( clipPlanes.isClipped( boxOutside ) );
    LVH_CHECK( !clipPlanes.isClipped( boxIntersect ) );
    LVH_CHECK( !clipPlanes.isEmpty() );
    clipPlanes.clear();
    LVH_CHECK( clipPlanes.isEmpty() );
    clipPlanes.reset();
    LVH_CHECK( !clipPlanes.isEmpty() );
    LVH_CHECK( !clipPlanes.isClipped( boxInside ) );
    LVH_CHECK( clipPlanes.isClipped( boxOutside ) );
    LVH_CHECK( !clipPlanes.isClipped( boxIntersect ) );
    return 0;
}

/* tests/lib/rendererParameters.cpp:25-44 (defaultValues) and :46-59 (copy) */
int lvh_selftest_renderer_parameters( void )
{
    const RendererParameters params;
    LVH_CHECK( params.getMaxLOD() == ( 4u << 1 ) + 1u ); /* NODEID_LEVEL_BITS = 4, livre/core/types.h:191 */
    LVH_CHECK( params.getMinLOD() == 0u );
    LVH_CHECK( !params.getSynchronousMode() );
    LVH_CHECK( params.getSamplesPerRay() == 0u );
    LVH_CHECK( params.getSamplesPerPixel() == 1u );
    LVH_CHECK( params.getSSE() == 4.0f );
    LVH_CHECK( params.getMaxGPUCacheMemoryMB() == 3072u );
    LVH_CHECK( params.getMaxCPUCacheMemoryMB() == 8192u );
    LVH_CHECK( !params.getRayLOD() ); /* the extension is off unless asked for */
    RendererParameters changed;
    changed.maxLOD = 42;
    const RendererParameters copy( changed );
    LVH_CHECK( copy.getMaxLOD() == 42u );
    RendererParameters assigned;
    assigned = changed;
    LVH_CHECK( assigned.getMaxLOD() == 42u );
    return 0;
#undef LVH_CHECK
}

int lvh_datasource_brick( const char* uri, uint64_t nodeId, uint8_t* out, size_t cap, size_t* n )
{
    try
    {
        DataSource dataSource{ std::string( uri ) };
        const ConstMemoryUnitPtr mem = dataSource.getData( NodeId( nodeId ) );
        if( !mem )
            return fail( "no data for node" );
        if( n ) *n = mem->getMemSize();
        if( out )
            std::memcpy( out, mem->getData< uint8_t >(), std::min( cap, mem->getMemSize() ) );
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

int lvh_datasource_info( const char* uri, uint32_t voxels[3], uint32_t maxBlock[3], uint32_t overlap[3],
                         float worldSize[3], uint32_t* depth, uint32_t rootBlocks[3], uint32_t* dataType,
                         uint32_t* compCount )
{
    try
    {
        DataSource dataSource{ std::string( uri ) };
        const VolumeInformation& vi = dataSource.getVolumeInfo();
        for( int a = 0; a < 3; ++a )
        {
            if( voxels ) voxels[a] = vi.voxels[a];
            if( maxBlock ) maxBlock[a] = vi.maximumBlockSize[a];
            if( overlap ) overlap[a] = vi.overlap[a];
            if( worldSize ) worldSize[a] = vi.worldSize[a];
            if( rootBlocks ) rootBlocks[a] = vi.rootNode.getBlockSize()[a];
        }
        if( depth ) *depth = vi.rootNode.getDepth();
        if( dataType ) *dataType = uint32_t( vi.dataType );
        if( compCount ) *compCount = vi.compCount;
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

int lvh_datasource_frame_range( const char* uri, uint32_t range[2] )
{
    try
    {
        if( !uri || !range ) return fail( "NULL argument" );
        DataSource dataSource{ std::string( uri ) };
        const Vector2ui r = dataSource.getVolumeInfo().frameRange;
        range[0] = r[0];
        range[1] = r[1];
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}

int lvh_datasource_node( const char* uri, uint64_t nodeId, int* valid, uint32_t blockSize[3],
                         uint32_t voxelBox[6], float worldBox[6] )
{
    try
    {
        DataSource dataSource{ std::string( uri ) };
        const LODNode node = dataSource.getNode( NodeId( nodeId ) );
        if( valid ) *valid = node.isValid() ? 1 : 0;
        for( int a = 0; a < 3; ++a )
        {
            if( blockSize ) blockSize[a] = node.getBlockSize()[a];
            if( voxelBox )
            {
                voxelBox[a] = node.getVoxelBox().getMin()[a];
                voxelBox[3 + a] = node.getVoxelBox().getMax()[a];
            }
            if( worldBox )
            {
                worldBox[a] = node.getWorldBox().getMin()[a];
                worldBox[3 + a] = node.getWorldBox().getMax()[a];
            }
        }
        return 0;
    }
    catch( const std::exception& e )
    {
        return fail( e.what() );
    }
}
}
