/* ThreadSanitizer run of the "hip" render pipeline's threading over the CPU stand-in of the device
 * ABI (vrc_stub.cpp): synchronous frames with a texture cache smaller than the visible set
 * (multi-pass, parallel loaders, the serial retry) and asynchronous frames with a moving camera
 * (background upload thread against the render thread, LRU eviction).  TEST INFRASTRUCTURE ONLY. */
#include <atomic>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "livre_hip/data.h"
#include "livre_hip_driver.h"

/* fail://#x,y,z,block : a regular tree whose bricks read fine until the N-th getData() of the process, which
 * throws from the loader thread that runs it (a corrupt brick, an I/O error).  The render call that waits for
 * the loaders -- or, in asynchronous mode, a later one -- must report it instead of waiting for ever. */
namespace
{
std::atomic< int > g_failAfter{ 1 << 30 };
class FailingDataSource : public livre::DataSourcePlugin
{
public:
    explicit FailingDataSource( const livre::DataSourcePluginData& d )
    {
        const livre::URI& uri = d.getURI();
        std::vector< std::string > par;
        const std::string frag = uri.getFragment();
        for( size_t a = 0; a <= frag.size(); )
        {
            const size_t b = std::min( frag.find( ',', a ), frag.size() );
            par.push_back( frag.substr( a, b - a ) );
            a = b + 1;
        }
        _volumeInfo.overlap = livre::Vector3ui( 4 );
        _volumeInfo.dataType = livre::DT_UINT8;
        _volumeInfo.compCount = 1;
        for( int a = 0; a < 3; ++a )
            _volumeInfo.voxels[a] = uint32_t( std::stoul( par.at( a ) ) );
        _volumeInfo.maximumBlockSize = livre::Vector3ui( uint32_t( std::stoul( par.at( 3 ) ) ) ) + _volumeInfo.overlap * 2u;
        _volumeInfo.frameRange = livre::FULL_FRAME_RANGE;
        if( !livre::fillRegularVolumeInfo( _volumeInfo ) )
            throw std::runtime_error( "Cannot setup the regular tree" );
    }
    livre::MemoryUnitPtr getData( const livre::LODNode& node ) final
    {
        if( --g_failAfter < 0 )
            throw std::runtime_error( "fail://: brick could not be read" );
        const livre::Vector3ui b = node.getBlockSize() + _volumeInfo.overlap * 2u;
        const size_t n = size_t( b[0] ) * b[1] * b[2];
        std::shared_ptr< livre::AllocMemoryUnit > m( new livre::AllocMemoryUnit( n ) );
        std::memset( m->getData< uint8_t >(), 40, n );
        return m;
    }
    static bool handles( const livre::DataSourcePluginData& d ) { return d.getURI().getScheme() == "fail"; }
};
livre::PluginRegisterer< FailingDataSource, const livre::DataSourcePluginData& > failRegisterer;
}

static int runFailing( int synchronous )
{
    lvh_params p;
    std::memset( &p, 0, sizeof( p ) );
    p.width = 48;
    p.height = 48;
    p.synchronous = synchronous;
    p.min_lod = p.max_lod = 2;
    p.gpu_cache_mb = 16;
    p.cpu_cache_mb = 4;
    lvh_app* app = nullptr;
    if( lvh_app_create( "fail://#64,64,64,16", "hip", &p, &app ) != 0 )
    {
        std::printf( "create failed: %s\n", lvh_last_error() );
        return 1;
    }
    g_failAfter = 20; /* 64 bricks: the failure comes while the loaders are busy */
    lvh_frame_stats st;
    int failedAt = -1;
    for( int i = 0; i < 50 && failedAt < 0; ++i )
    {
        if( lvh_app_render_frame( app, nullptr, &st ) != 0 )
            failedAt = i;
        if( !synchronous )
            lvh_app_wait_uploads( app );
    }
    const bool told = failedAt >= 0 && std::strstr( lvh_last_error(), "brick could not be read" ) != nullptr;
    std::printf( "failing source, sync %d: render call %d reported \"%s\"\n", synchronous, failedAt,
                 failedAt >= 0 ? lvh_last_error() : "nothing" );
    /* and the pipeline is still usable: the source recovers, the next frames complete */
    g_failAfter = 1 << 30;
    int ok = 0;
    for( int i = 0; i < 4; ++i )
    {
        ok += lvh_app_render_frame( app, nullptr, &st ) == 0;
        lvh_app_wait_uploads( app );
    }
    lvh_app_destroy( app );
    return ( told && ok >= 3 && st.n_available == 64 ) ? 0 : 1;
}

static int run( int synchronous, uint32_t gpuMb, int frames, int rayLod = 0 )
{
    lvh_params p;
    std::memset( &p, 0, sizeof( p ) );
    p.width = 64;
    p.height = 64;
    p.synchronous = synchronous;
    p.min_lod = rayLod ? 0 : 3; /* per-ray LOD: the whole hierarchy above the cut */
    p.max_lod = 3;
    p.sse = rayLod ? 0.5f : 0.f;
    p.gpu_cache_mb = gpuMb;
    p.cpu_cache_mb = 4;
    lvh_app* app = nullptr;
    if( lvh_app_create( "mem://#128,128,128,16", "hip", &p, &app ) != 0 )
    {
        std::printf( "create failed: %s\n", lvh_last_error() );
        return 1;
    }
    lvh_frame_stats st;
    lvh_app_set_ray_lod( app, rayLod );
    const float pos[3] = { 0.f, 0.f, 1.5f }, look[3] = { 0.f, 0.f, 0.f };
    for( int i = 0; i < frames; ++i )
    {
        /* every view twice in a row (the repeated frame takes the kept brick list), small and large moves */
        lvh_app_set_camera( app, pos, look, 0.3f + 0.05f * float( ( i / 2 ) % 7 ) + ( i % 11 == 0 ? 1.5f : 0.f ), 0.2f );
        if( lvh_app_render_frame( app, nullptr, &st ) != 0 )
        {
            std::printf( "render failed: %s\n", lvh_last_error() );
            return 1;
        }
        if( !synchronous && ( i % 5 ) == 4 )
            lvh_app_wait_uploads( app );
    }
    lvh_app_wait_uploads( app );
    std::printf( "sync %d cache %u MB: available %llu not available %llu passes %u\n", synchronous, gpuMb,
                 (unsigned long long)st.n_available, (unsigned long long)st.n_not_available, st.n_passes );
    lvh_app_destroy( app );
    return 0;
}

int main()
{
    int rc = run( 1, 16, 6 ); /* everything fits: one pass, 512 bricks through the loaders */
    rc |= run( 1, 2, 4 );     /* 151 slots for 512 bricks: four passes per frame */
    rc |= run( 0, 16, 30 );   /* asynchronous, fits */
    rc |= run( 0, 2, 60 );    /* asynchronous under cache pressure */
    rc |= run( 1, 16, 6, 1 ); /* per-ray LOD: cut + ancestors, synchronous */
    rc |= run( 0, 16, 30, 1 ); /* per-ray LOD, asynchronous: render the resident part of the hierarchy */
    rc |= run( 0, 2, 30, 1 );  /* hierarchy larger than the atlas: falls back to the per-brick cut */
    rc |= runFailing( 1 );     /* a loader thread throws: the waiting render call reports it (no hang) */
    rc |= runFailing( 0 );     /* the asynchronous upload pipeline throws: a later render call reports it */
    if( rc == 0 )
        std::printf( "DONE\n" );
    return rc;
}
