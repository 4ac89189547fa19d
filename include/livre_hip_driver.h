/*
 * livre_hip_driver.h -- flat C entry points of libLivreHipRaycastPipeline.so for hosts that are
 * not C++ (the Python bench and tests bind them with ctypes).  It is the headless stand-in for
 * what apps/livre + livre/eq do around the plugin: open the data source (livre/eq/Node.cpp:51-77),
 * create the RenderPipeline by renderer name (livre/eq/Window.cpp:59-63), build RenderInputs per
 * frame and call RenderPipeline::render (livre/eq/Channel.cpp:259-308).  Everything below the
 * call is the C++ plugin surface (libre_amd/host) and the device C ABI (vrc_hip.h).
 */
#ifndef LIVRE_HIP_DRIVER_H
#define LIVRE_HIP_DRIVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lvh_app lvh_app;

/* rendererParameters.fbs:4-13 + ApplicationParameters.cpp:52-61 defaults when a field is 0 /
 * negative as documented */
typedef struct
{
    int device;                 /* HIP device of this process */
    uint32_t width, height;     /* full frame */
    uint32_t tile[4];           /* x, y, w, h of this process' sort-first tile; w = 0 -> full frame */
    int synchronous;            /* --synchronous */
    uint32_t samples_per_ray;   /* --samples-per-ray, 0 = auto */
    uint32_t min_lod, max_lod;  /* --min-lod / --max-lod; max_lod 0 -> 9 (default) */
    float sse;                  /* --sse, 0 -> 4.0 */
    uint32_t gpu_cache_mb;      /* --gpu-cache-mem, 0 -> 3072 */
    uint32_t cpu_cache_mb;      /* --cpu-cache-mem, 0 -> 8192 */
} lvh_params;

typedef struct
{
    uint64_t n_available, n_not_available, n_render_available; /* RenderStatistics */
    uint32_t n_passes;
    float kernel_ms;   /* HIP-event time of the last raycast kernel */
    uint64_t samples;  /* composited samples of the last kernel when counting is on */
    uint32_t samples_per_ray;
    double kernel_ms_sum;     /* raycast kernel time summed over the launches since the last */
    uint32_t kernel_launches; /* lvh_app_get_stats call (HIP events on the render stream)   */
    uint32_t ray_lod;         /* 1: the frame was rendered with per-ray LOD (lvh_app_set_ray_lod) */
} lvh_frame_stats;

const char* lvh_last_error( void );
/* volume_uri: mem://#x,y,z,block | raw://file.raw#x,y,z,type | hash://#x,y,z,block ;
 * renderer: "hip" (PluginFactory: unknown name -> error "No plugin implementation available") */
int lvh_app_create( const char* volume_uri, const char* renderer, const lvh_params* params,
                    lvh_app** out );
void lvh_app_destroy( lvh_app* app );
/* camera: ApplicationParameters camera-position / camera-lookat, CameraSettings::spinModel */
int lvh_app_set_camera( lvh_app* app, const float position[3], const float lookat[3],
                        float spin_x, float spin_y );
int lvh_app_set_modelview( lvh_app* app, const float mv[16] );
/* FrameInfo::timeStep of the following frames (the frame number of livre/eq/Channel.cpp:259-270): must lie in the
 * data source's frame range (lvh_datasource_frame_range) */
int lvh_app_set_time_step( lvh_app* app, uint32_t time_step );
int lvh_app_set_colormap( lvh_app* app, const float rgba256[1024] );
int lvh_app_set_clip_planes( lvh_app* app, const float* planes, uint32_t n );
/* sort-first row bands rendered by this process in one launch: bands (y0[i], h[i]) of the full
 * frame, stacked in that order in its pixel buffer (replaces params.tile; n = 0 -> back to it) */
int lvh_app_set_bands( lvh_app* app, const uint32_t* y0, const uint32_t* h, uint32_t n );
int lvh_app_set_option( lvh_app* app, int vrc_option, int64_t value );
/* RenderInputs::dataSourceRange for volumes that are not uint8 (the reference forces (0,255),
 * livre/eq/Channel.cpp:284, and its CUDA renderer ignores the field; 16-bit volumes are an
 * extension here and default to (0,65535)) */
int lvh_app_set_data_range( lvh_app* app, float lo, float hi );
/* EXTENSION (BASELINE C5): per-ray adaptive LOD.  The pipeline makes the ancestors of the visible
 * set (SelectVisibles cut at --sse) resident as well and every ray applies the screen-space-error
 * rule where it is (vrc_set_ray_lod).  Frames whose hierarchy does not fit the atlas in one pass
 * are rendered with the per-brick cut (stats.ray_lod = 0). */
int lvh_app_set_ray_lod( lvh_app* app, int enable );
/* Frames in flight: the application keeps n Renderer("hip") instances (each with its own device
 * context, stream and pixel buffer) over ONE pipeline (one brick atlas, one pair of caches), as
 * RenderPipelinePlugin::render( Renderer&, ... ) allows; Equalizer's default latency of one frame
 * (livre/eq/Client.cpp:210-237 frame loop) corresponds to n = 2.  select_slot chooses the instance
 * the following set_stream / set_framebuffer / set_option / render_frame / get_stats calls use. */
int lvh_app_set_frames_in_flight( lvh_app* app, uint32_t n );
int lvh_app_select_slot( lvh_app* app, uint32_t slot );
/* render on a caller-owned stream / into caller-owned device memory (tile gather) */
int lvh_app_set_stream( lvh_app* app, void* hip_stream );
int lvh_app_set_framebuffer( lvh_app* app, void* device_rgba );
/* one frame: Channel::frameDraw. host_rgba may be NULL (frame stays in HBM). */
int lvh_app_render_frame( lvh_app* app, float* host_rgba, lvh_frame_stats* stats );
int lvh_app_get_stats( lvh_app* app, lvh_frame_stats* stats ); /* kernel_ms/samples after sync */
int lvh_app_wait_uploads( lvh_app* app );
int lvh_app_synchronize( lvh_app* app );
/* introspection used by the parity tests */
int lvh_app_volume_info( lvh_app* app, uint32_t voxels[3], uint32_t max_block[3],
                         uint32_t overlap[3], float world_size[3], uint32_t* depth,
                         uint32_t root_blocks[3] );
int lvh_app_visible_set( lvh_app* app, uint64_t* ids, size_t capacity, size_t* n );
/* ---- sort-first tile exchange (include/vrc_hip.h: vrc_comm_*, vrc_gather_tiles) driven from the host:
 * what eq::Compositor::assembleFrame does for the reference (livre/eq/Channel.cpp:519-523).  One rank calls
 * lvh_comm_unique_id and hands the 128 bytes to all; every rank then creates its communicator (collective),
 * declares the frame's band layout (the same list on all ranks; a rank's own bands are those of
 * lvh_app_set_bands, in list order) and, per frame or batch of frames, calls lvh_app_gather_tiles with its
 * stacked bands; on `root` the bands land at their rows of `frame_device`. */
int lvh_comm_unique_id( uint8_t id[128] );
int lvh_app_comm_create( lvh_app* app, int rank, int world, const uint8_t* id );
int lvh_app_set_layout( lvh_app* app, const uint32_t* rank, const uint32_t* y0, const uint32_t* h, uint32_t n );
int lvh_app_gather_tiles( lvh_app* app, uint32_t n_frames, const void* local_device, size_t local_frame_stride,
                          void* frame_device, size_t frame_stride, int root, void* hip_stream );
/* ids of the bricks of the last (pass of the last) frame in the order the renderer handed them to the device
 * layer: front to back by box-centre distance (CudaRaycastRenderer.cpp:160-163).  Bricks at (nearly) equal
 * distance come in an order the reference leaves to std::sort and to the rounding of vmmlib's transform;
 * tests hand this list to the oracle instead of re-deriving it. */
int lvh_app_node_order( lvh_app* app, uint64_t* ids, size_t cap, size_t* n );
int lvh_app_view_matrices( lvh_app* app, float mv[16], float proj[16] );
int lvh_app_cache_stats( lvh_app* app, uint64_t tex[4], uint64_t data[4] ); /* used, max, count, misses */
/* standalone LOD cut with explicit matrices (tests/lib/lodSelection.cpp harness) */
int lvh_select_visibles( const char* volume_uri, const float mv[16], const float proj[16],
                         uint32_t window_height, float sse, uint32_t min_lod, uint32_t max_lod,
                         uint64_t* ids, size_t capacity, size_t* n );
/* host-only checks of the mirrored classes (no GPU): returns 0 when all pass, else the number
 * of the first failing check; message via lvh_last_error */
int lvh_selftest_cache( void );
int lvh_selftest_plugin_factory( void );
int lvh_selftest_camera( float out_matrices[4][16] );
int lvh_selftest_clip_planes( void );         /* tests/core/clipPlanes.cpp:29-59; 0 or the failing line */
int lvh_selftest_renderer_parameters( void ); /* tests/lib/rendererParameters.cpp:25-59 */
int lvh_datasource_brick( const char* volume_uri, uint64_t node_id, uint8_t* out, size_t capacity,
                          size_t* n );

/* data-source metadata without a renderer (no GPU): VolumeInformation and one LODNode.
 * data_type is the DataType enum of livre/core/data/VolumeInformation.h (DT_UINT8 = 1 ...). */
int lvh_datasource_info( const char* volume_uri, uint32_t voxels[3], uint32_t max_block[3],
                         uint32_t overlap[3], float world_size[3], uint32_t* depth,
                         uint32_t root_blocks[3], uint32_t* data_type, uint32_t* comp_count );
/* VolumeInformation::frameRange: [first, end) of the time steps the source holds (livre/core/data/
 * VolumeInformation.h; uvf://: one per TOC block of the file, datasources/uvf/UVFDataSource.cpp:144) */
int lvh_datasource_frame_range( const char* volume_uri, uint32_t range[2] );
int lvh_datasource_node( const char* volume_uri, uint64_t node_id, int* valid, uint32_t block_size[3],
                         uint32_t voxel_box[6], float world_box[6] );

#ifdef __cplusplus
}
#endif
#endif
