/* ThreadSanitizer run of the "hip" render pipeline's threading over the CPU stand-in of the device
 * ABI (vrc_stub.cpp): synchronous frames with a texture cache smaller than the visible set
 * (multi-pass, parallel loaders, the serial retry) and asynchronous frames with a moving camera
 * (background upload thread against the render thread, LRU eviction).  TEST INFRASTRUCTURE ONLY. */
#include <cstdio>
#include <cstring>

#include "livre_hip_driver.h"

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
    if( rc == 0 )
        std::printf( "DONE\n" );
    return rc;
}
