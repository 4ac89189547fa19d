/*
 * vrc_hip.h -- C ABI of libvrc_hip.so, the MI355X (gfx950) device layer of the volume
 * raycaster.  It replaces, entry point for entry point, the reference's CUDA device layer
 * renderers/cudaRaycaster/cuda/ (classes cuda::Renderer, cuda::TexturePool, cuda::ColorMap,
 * cuda::ClipPlanes, cuda::PixelBufferObject), which the host plugin classes
 * CudaRaycastRenderer / CudaTexturePool / CudaTextureObject call.  Everything here is
 * plain C: opaque handles, PODs, pointers and sizes.  No torch, no C++ types.
 *
 * Reference citations are path:line relative to the reference root.
 *
 * Return convention: 0 = VRC_OK, otherwise a VRC_E* code; vrc_last_error() returns the
 * thread-local message (the reference throws std::runtime_error from checkCudaErrors,
 * cuda/cuda.h:39-53; the C++ shim rethrows the same exception types).
 *
 * Threading (mirrors SURVEY 8b): vrc_pool_* are thread-safe (the reference guards its free
 * list with a mutex, cuda/TexturePool.cu:179-185, and calls copyToSlot from 3 threads);
 * vrc_update/pre_render/render/post_render are single-threaded per context.  Unlike the
 * reference (quirk Q9) every upload is ordered before the next vrc_render by an event.
 */
#ifndef VRC_HIP_H
#define VRC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRC_OK 0
#define VRC_EINVAL 1     /* bad argument */
#define VRC_EHIP 2       /* a HIP runtime call failed (message has hipGetErrorString) */
#define VRC_EFULL 3      /* no free slot in the pool: slot = (-1,-1,-1), TexturePool.cu:180-181 */
#define VRC_ENOMEM 4     /* allocation failed */
#define VRC_EUNSUPPORTED 5 /* unsupported data type / channel count, TexturePool.cu:66-67 */
#define VRC_EHIERARCHY 6  /* vrc_set_ray_lod is on and the node list of vrc_render is not a brick hierarchy
                          * (nothing was rendered; the caller may render its per-brick cut instead) */
#define VRC_ECOMM 7       /* RCCL missing or an RCCL call failed (vrc_comm_*, vrc_gather_tiles) */

typedef struct vrc_ctx vrc_ctx;   /* replaces cuda::Renderer (cuda/Renderer.cuh:69-112) */
typedef struct vrc_pool vrc_pool; /* replaces cuda::TexturePool (cuda/TexturePool.cuh:42-103) */

/* cuda/Renderer.cuh:35-41 -- 12 floats, 48 bytes, same field order */
typedef struct
{
    float textureMin[3];  /* normalized atlas origin of the brick interior */
    float textureSize[3]; /* normalized atlas size of the brick interior */
    float aabbMin[3];     /* world box min */
    float aabbSize[3];    /* world box size */
} vrc_node_data;

/* cuda/Renderer.cuh:46-56, same field order.  Matrices are column-major float[16]. */
typedef struct
{
    float eyePosition[3];
    uint32_t glViewport[4]; /* x, y, w, h; the kernel maps pixel (px,py) of the buffer */
    float invProjMatrix[16];
    float modelViewMatrix[16]; /* carried for ABI shape; unused by the kernel (quirk Q3) */
    float invViewMatrix[16];
    float aabbMin[3];
    float aabbMax[3];
    float nearPlane;
} vrc_view_data;

/* cuda/Renderer.cuh:59-66, same field order */
typedef struct
{
    uint32_t samplesPerRay;
    uint32_t samplesPerPixel;  /* unused by the kernel (quirk Q3) */
    uint32_t maxSamplesPerRay; /* opacity-correction reference, 32 in the reference */
    uint32_t datatype;         /* unused by the kernel (quirk Q2) */
    float dataSourceRange[2];
} vrc_render_data;

/* per-render statistics (not in the reference; feeds bench.py's Msamples/s and roofline) */
typedef struct
{
    float kernel_ms;        /* HIP-event time of the last raycast kernel on the ctx stream */
    uint64_t samples;       /* samples composited by the last vrc_render (0 if counting is off) */
    uint32_t kernel_variant; /* which kernel ran: VRC_KERNEL_* */
    uint32_t grid_dims[3];  /* brick-grid dims used by the DDA kernel (0 if not used) */
    double kernel_ms_sum;   /* sum of HIP-event times of the raycast kernels launched since ... */
    uint32_t kernel_launches; /* ... the previous vrc_get_stats call, and how many they were */
} vrc_stats;

/* ---- options (vrc_set_option) ---------------------------------------------------------- */
#define VRC_OPT_KERNEL 1          /* VRC_KERNEL_AUTO (default) | _REFERENCE_ORDER | _GRID_DDA | _LDS | _PACKED */
#define VRC_OPT_FILTER 2          /* VRC_FILTER_NEAREST (reference parity, default) | VRC_FILTER_TRILINEAR */
#define VRC_OPT_TF_FRAC_BITS 3    /* 8 (default, CUDA 1.8 fixed-point lerp weight) | 0 exact float */
#define VRC_OPT_COUNT_SAMPLES 4   /* 0 (default) | 1: count composited samples (slower kernel) */
#define VRC_OPT_TILE_ORDER 5      /* 1 (default): heaviest-first tile schedule | 0: row-major tiles */
#define VRC_OPT_STEPPING 6        /* sample positions inside a brick: 1 (default) 8.24 fixed-point
                                   * voxel-space increments | 0 the reference's float world-space
                                   * accumulation (cuda/Renderer.cu:208: pos += step) */

#define VRC_OPT_VARIANT 7         /* which of the reference's two raycasters the frame must match:
                                   * VRC_VARIANT_CUDARAYCASTER (default) | VRC_VARIANT_GLRAYCASTER */

#define VRC_OPT_KERNEL_USED 8      /* read-only (vrc_get_option): VRC_KERNEL_* of the last vrc_render, without the
                                   * synchronisation vrc_get_stats implies.  Every kernel but REFERENCE_ORDER
                                   * finds the bricks of a ray through the brick grid: the order of the node
                                   * list does not matter to it */

#define VRC_OPT_KERNEL_TIMING 9    /* 1 (default): a HIP event pair around every raycast launch feeds vrc_get_stats'
                                   * kernel times; 0: no events (two host calls less per vrc_render; vrc_get_stats
                                   * then reports 0 ms and 0 launches) */

#define VRC_OPT_DEPTH_SPLIT 10     /* 0 (default) | 1: two waves per 8x8 tile, one for the bricks in the near and one for
                                   * those in the far half of every ray, composited with `over`: halves the
                                   * latency of launches too small to fill the GPU (a rank's share of a sort-first
                                   * frame) at ~15 % more work.  Same samples (every brick is marched whole by one of
                                   * the two); taken only where exact: frames in which early ray termination cannot
                                   * occur (largest classified opacity ^ most samples per ray), first pass of a
                                   * frame, the point-sampling grid-walk kernel; silently the plain kernel else */

#define VRC_OPT_ERT_COMPACTION 11  /* 0 (default) | 2..8 = P: ray compaction for early ray termination
                                   * (cuda/Renderer.cu:219-226).  The march runs in P launches, one per slab of the
                                   * brick grid along the view axis; after each, a wave-level ballot packs the rays
                                   * that are still below the opacity threshold into a list, and the next launch marches
                                   * 64 live rays per wave from it instead of tiles whose lanes have mostly finished.
                                   * Same bricks, same order, same arithmetic per ray: the frame is bit-identical to
                                   * the single launch.  Pays only for frames in which many, but not all, rays of a
                                   * tile end early (DESIGN.md section 4 has the measurements); the point-sampling
                                   * grid-walk kernel only, silently the plain kernel else.  With per-launch lists:
                                   * vrc_get_ray_counts */

#define VRC_OPT_GREY_TABLE 12      /* 1 (default) | 0.  A transfer function whose red, green and blue are equal in every
                                   * entry (bit for bit) lets the kernels keep (grey, alpha)
                                   * instead of four floats per classified-table entry and per colour: half the
                                   * table bytes in LDS and three fused multiply-adds less per sample.  The three colour
                                   * channels of the reference's blend (cuda/Renderer.cu:83-93) are then the same
                                   * operations on the same numbers, so the frame is bit-identical; first pass of
                                   * a frame only (the pixel starts from zero).  0: always the four-float form */

#define VRC_OPT_PACKED_ATLAS 13    /* 1 (default) | 0.  May VRC_KERNEL_AUTO build the pool's tap-packed atlas (VRC_KERNEL_PACKED:
                                   * 2.25 times the bytes of the brick atlas, on top of the budget given to
                                   * vrc_pool_create) the first time a frame with the trilinear filter could use it?
                                   * 0: AUTO stays with the LDS-staged / gather forms; asking for VRC_KERNEL_PACKED
                                   * explicitly still builds it */

#define VRC_VARIANT_CUDARAYCASTER 0 /* renderers/cudaRaycaster/cuda/Renderer.cu:95-230 */
#define VRC_VARIANT_GLRAYCASTER 1   /* renderers/glRaycaster/shaders/fragRaycast.glsl:113-215: pixel centre
                                     * +0.5, hit test t0 <= t1, first sample of a brick snapped to the
                                     * global step lattice, clip planes per brick after the snap (the
                                     * RGBA8 transfer function of GLRaycastRenderer.cpp:188-192 is the
                                     * caller's: pass the quantised table to vrc_update) */

#define VRC_FILTER_NEAREST 0   /* cuda/TexturePool.cu:167 (cudaFilterModePoint): the reference */
#define VRC_FILTER_TRILINEAR 1 /* extension: texel centres at i+0.5, float weights, transfer function
                                * and opacity correction evaluated per sample on the interpolated density */

#define VRC_KERNEL_AUTO 0            /* bricks of one size: GRID_DDA (meets them in the reference's order, sample for
                                      * sample); bricks of mixed sizes (an LOD cut): REFERENCE_ORDER up to
                                      * VRC_REFERENCE_ORDER_MAX_NODES bricks -- the reference composites in the host's
                                      * centre-distance order, which is not a visibility order for every ray once
                                      * brick sizes differ (cuda/Renderer.cu:172-199) -- above that GRID_DDA */
#define VRC_REFERENCE_ORDER_MAX_NODES 4096
#define VRC_KERNEL_REFERENCE_ORDER 1 /* O(nodes) loop per ray in host order, cuda/Renderer.cu:172-227 */
#define VRC_KERNEL_GRID_DDA 2        /* 3-D DDA over the brick grid; needs a grid-aligned node set */
#define VRC_KERNEL_LDS 3             /* grid DDA + voxels staged through LDS per wave and round (needs
                                      * overlap >= 1); what AUTO picks for the trilinear filter where the
                                      * tap-packed atlas (VRC_KERNEL_PACKED) is not available */
#define VRC_KERNEL_PACKED 5          /* the trilinear filter through the pool's tap-packed atlas: a second atlas, 2.25 times
                                      * the bytes, whose texel at (x,y,z) holds a voxel and its neighbour along z,
                                      * v[x,y,z] | v[x,y,z+1] << 8 (16-bit voxels: << 16), laid out so that the texels at
                                      * x and x + 1 are always neighbours in memory; allocated and filled on first use
                                      * and kept up to date by every later upload: the eight taps of a sample are TWO
                                      * gathers (rows y and y + 1; 4 bytes each, 8 for 16-bit voxels).  Needs
                                      * VRC_FILTER_TRILINEAR, 8- or 16-bit bricks with overlap >= 1 in slots of at
                                      * most 248 voxels a side, VRC_OPT_TF_FRAC_BITS = 8, VRC_OPT_STEPPING = 1
                                      * (VRC_EINVAL otherwise) and the device memory (VRC_ENOMEM).  Same sample
                                      * positions, weights and arithmetic as the LDS-staged form: the same frame,
                                      * bit for bit.  What AUTO picks for the trilinear filter where all of this
                                      * holds (VRC_OPT_PACKED_ATLAS) */
#define VRC_KERNEL_RAY_LOD 4         /* reported by vrc_get_stats when vrc_set_ray_lod is on; not selectable.  Under
                                      * per-ray LOD VRC_OPT_KERNEL chooses how the hierarchy walk takes its samples:
                                      * AUTO = for the trilinear filter the tap-packed atlas where VRC_KERNEL_PACKED
                                      * applies, else staged through LDS (8- or 16-bit bricks, overlap >= 1,
                                      * VRC_OPT_TF_FRAC_BITS 8), by gathers otherwise; GRID_DDA = gathers; LDS = staged or
                                      * VRC_EINVAL; PACKED = the packed atlas or VRC_EINVAL; vrc_last_kernel names the
                                      * instance that ran */

/* ---- context ---------------------------------------------------------------------------- */
/* cuda::Renderer::Renderer() (cuda/Renderer.cu:234-238); device is explicit (fixes Q11) */
int vrc_ctx_create( int device, vrc_ctx** out );
void vrc_ctx_destroy( vrc_ctx* ctx );
/* launch on a caller-owned hipStream_t (pass the handle as void*); NULL = the ctx's own stream */
int vrc_ctx_set_stream( vrc_ctx* ctx, void* hip_stream );
int vrc_set_option( vrc_ctx* ctx, int option, int64_t value );
int vrc_get_option( vrc_ctx* ctx, int option, int64_t* value );
/* EXTENSION (BASELINE C5): per-ray adaptive LOD.  The reference selects the LOD per brick on the host
 * (livre/core/render/SelectVisibles.cpp:52-68: a brick is fine enough when
 * worldSpacePerVoxel / worldSpacePerPixel * near / (near + distance) <= screenSpaceError at the
 * point of its box nearest to the near plane).  With this on, the node list of vrc_render is a
 * hierarchy of resident bricks (a cut of the octree plus any of its ancestors): boxes of different
 * levels nest; every level is a regular grid of bricks anchored at the min corner of all boxes
 * (border bricks may be smaller, levels need not align with each other: UVF trees), one brick per
 * cell at most; else VRC_EHIERARCHY.  Every ray applies the criterion where it enters a brick: it
 * samples the coarsest level that is fine enough there (else the next coarser one present, else
 * the next finer), with step and opacity exponent scaled by 2^level.  world_space_per_pixel =
 * (frustum.top - frustum.bottom) / window height, as in SelectVisibles.cpp:57.  cudaRaycaster
 * variant only. */
int vrc_set_ray_lod( vrc_ctx* ctx, int enable, float screen_space_error, float world_space_per_pixel );

/* ---- texture pool (brick atlas) ----------------------------------------------------------- */
/* cuda::TexturePool::TexturePool (cuda/TexturePool.cu:101-173).  max_block is the slot size
 * in voxels (block + 2*overlap), max_bytes the atlas budget.  Slot grid and free-list order
 * follow TexturePool.cu:128-144 with VRC_MAX_TEXTURE_3D standing in for maxTexture3D. */
#define VRC_MAX_TEXTURE_3D 4096
int vrc_pool_create( vrc_ctx* ctx, size_t bytes_per_voxel, int is_signed, int is_float,
                     size_t n_components, const uint32_t max_block[3], size_t max_bytes,
                     vrc_pool** out );
void vrc_pool_destroy( vrc_pool* pool );
/* cuda::TexturePool::copyToSlot (cuda/TexturePool.cu:175-203): host brick of size[] voxels,
 * tightly packed, x fastest.  Writes the normalized slot origin; on a full pool returns
 * VRC_EFULL and writes (-1,-1,-1).  The host pointer is only borrowed for the call. */
int vrc_pool_copy_to_slot( vrc_pool* pool, const void* host_brick, const uint32_t size[3],
                           float slot_out[3] );
/* same, but the brick already lives in device memory (row-major, tightly packed) */
int vrc_pool_copy_to_slot_device( vrc_pool* pool, const void* device_brick,
                                  const uint32_t size[3], float slot_out[3] );
/* cuda::TexturePool::releaseSlot (cuda/TexturePool.cu:210-214) */
int vrc_pool_release_slot( vrc_pool* pool, const float slot[3] );
/* getSlotMemSize / getTextureSize / getTextureMem (cuda/TexturePool.cuh:80-95) + free count */
int vrc_pool_info( const vrc_pool* pool, size_t* slot_bytes, uint32_t atlas_dim[3],
                   size_t* atlas_bytes, uint32_t slots[3], uint32_t* free_slots );
/* block until every pending upload of the pool has landed in HBM */
int vrc_pool_synchronize( vrc_pool* pool );
/* debug/test: read back the voxel at logical atlas coordinate (x,y,z) region into host memory,
 * row-major; used by the parity tests to check the atlas layout transform */
int vrc_pool_read_region( vrc_pool* pool, const uint32_t origin[3], const uint32_t size[3],
                          void* host_out );

/* Histogram of a resident brick as a side kernel (the reference bins the CPU copy,
 * livre/lib/cache/HistogramObject.cpp:36-119): voxels [origin, origin+size) of the slot in
 * slot-local coordinates (origin = overlap, size = the node's voxel box: the interior, :94-97);
 * integral voxels are binned over the type's range, bin = v / (range / bin_count) (:104-110);
 * every voxel adds scale_factor (8^(depth-1-level), :158-162).  bin_count must divide the range
 * (256 for uint8, 1024 for uint16 in the reference, :167-176).  Synchronous; host_bins receives
 * bin_count values. */
int vrc_pool_histogram( vrc_pool* pool, const float slot[3], const uint32_t origin[3],
                        const uint32_t size[3], uint32_t bin_count, uint64_t scale_factor,
                        uint64_t* host_bins );

/* ---- renderer ----------------------------------------------------------------------------- */
/* cuda::Renderer::update (cuda/Renderer.cu:245-250): 256 RGBA float texels as
 * lexis ColorMap::sampleColors<float>(256,0,256,0) yields them (cuda/ColorMap.cu:56-65), and
 * up to 6 clip planes (nx,ny,nz,d) (cuda/ClipPlanes.cu:32-46).  n_planes == 0 clears the
 * planes (the reference keeps stale ones, an obvious slip). */
int vrc_update( vrc_ctx* ctx, const float tf_rgba[256 * 4], const float* planes, uint32_t n_planes );
/* cuda::Renderer::preRender (cuda/Renderer.cu:252-257) + PixelBufferObject::resize/mapBuffer
 * (cuda/PixelBufferObject.cu:43-81): (re)allocate W x H float4 and clear it to 0.
 * W = glViewport[2], H = glViewport[3] (the reference's w-x / h-y is quirk Q10). */
int vrc_pre_render( vrc_ctx* ctx, const vrc_view_data* view );
/* render into caller-owned device memory instead (the reference renders into a GL-owned PBO):
 * width*height float4, cleared by vrc_pre_render like the internal one.  NULL returns to the
 * internal buffer. */
int vrc_set_framebuffer( vrc_ctx* ctx, void* device_rgba, uint32_t width, uint32_t height );
int vrc_get_framebuffer( vrc_ctx* ctx, void** device_rgba, uint32_t* width, uint32_t* height );
/* Sort-first row bands in ONE launch: the pixel buffer holds n_rows rows, row i of it being
 * frame row frame_rows[i] of the frame described by glViewport (whose w,h stay the FULL frame).
 * Takes effect at the next vrc_pre_render (buffer = glViewport.w x n_rows).  n_rows = 0 returns
 * to the plain case (buffer = glViewport.w x glViewport.h).  Rays are those of the full frame,
 * bit for bit.  (Replaces Equalizer's per-channel pixel viewport, livre/eq/Channel.cpp:272-290,
 * for a non-contiguous set of rows.) */
int vrc_set_row_map( vrc_ctx* ctx, const uint32_t* frame_rows, uint32_t n_rows );
/* cuda::Renderer::render (cuda/Renderer.cu:274-297): node table H2D + one rayCast pass that
 * accumulates into the pixel buffer.  nodes are in the host's front-to-back order. */
int vrc_render( vrc_ctx* ctx, const vrc_view_data* view, const vrc_node_data* nodes,
                uint32_t n_nodes, const vrc_render_data* render, vrc_pool* pool );
/* cuda::Renderer::postRender (cuda/Renderer.cu:299-326): the reference unmaps the PBO and
 * glDrawPixels it; here the frame is complete on the stream and, if host_rgba is non-NULL,
 * copied to W*H*4 floats of host memory (synchronous). */
int vrc_post_render( vrc_ctx* ctx, float* host_rgba );
int vrc_synchronize( vrc_ctx* ctx );
int vrc_get_stats( vrc_ctx* ctx, vrc_stats* out );
/* Ray compaction (VRC_OPT_ERT_COMPACTION = P) of the last vrc_render: counts[p] = rays still alive after launch p
 * (p = 0..P-2: the length of the list launch p + 1 marched), 0 for the rest of the 8 entries; *parts = P, or 0 when
 * that render did not use compaction.  Waits for the render. */
int vrc_get_ray_counts( vrc_ctx* ctx, uint32_t counts[8], int* parts );

/* ---- sort-first tile exchange (multi-GPU) ------------------------------------------------------- */
/* One process per GPU renders row bands of the frame (vrc_set_row_map); the display rank receives
 * them over RCCL (xGMI inside a node) directly at their rows of the full frame.  This is the step
 * eq::Compositor::assembleFrame performs for the reference's sort-first compounds
 * (livre/eq/Channel.cpp:519-523; tiles: livre/eq/Channel.cpp:272-290) for a host without Equalizer.
 * No brick data moves between ranks.  RCCL is bound at run time (librccl.so.1); without it every
 * call below returns VRC_ECOMM except for a world of one rank. */
typedef struct vrc_comm vrc_comm;
#define VRC_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
/* ncclGetUniqueId: called by ONE rank; the caller hands the bytes to every rank (any transport) */
int vrc_comm_unique_id( uint8_t id_out[VRC_COMM_ID_BYTES] );
/* ncclCommInitRank on ctx's device; collective over the `world` ranks.  world == 1 needs no id
 * (may be NULL) and no RCCL. */
int vrc_comm_create( vrc_ctx* ctx, int rank, int world, const uint8_t id[VRC_COMM_ID_BYTES], vrc_comm** out );
void vrc_comm_destroy( vrc_comm* comm );
int vrc_comm_info( const vrc_comm* comm, int* rank, int* world );
/* one row band of the frame: rows [frame_row, frame_row + rows) are rendered by `rank` */
typedef struct
{
    uint32_t rank;
    uint32_t frame_row;
    uint32_t rows;
} vrc_band;
/* Collective over the communicator, asynchronous on hip_stream (NULL: ctx's render stream, i.e. behind
 * the vrc_render calls that produced the bands; another stream is made to wait for what ctx's render
 * stream holds at the time of the call).  `bands` lists every band of the width x height frame, identical
 * on all ranks; a rank's bands lie stacked in `local` in list order (what vrc_set_row_map renders),
 * width x rows RGBA32F each.  On rank `root` band b lands at row bands[b].frame_row of `frame` (its
 * own bands by device copies); `frame` is ignored elsewhere.  n_frames > 1 moves that many consecutive
 * frames with one group of sends/receives: frame f of a rank starts local_frame_stride bytes after
 * frame f-1 in `local`, and frame_stride bytes in `frame`.  Every band must lie inside the frame
 * (frame_row + rows <= height, checked in 64 bits) and a frame stride must hold a frame: nothing is
 * queued on any rank otherwise (VRC_EINVAL). */
int vrc_gather_tiles( vrc_ctx* ctx, vrc_comm* comm, const vrc_band* bands, uint32_t n_bands, uint32_t width,
                      uint32_t height, uint32_t n_frames, const void* local_device, size_t local_frame_stride,
                      void* frame_device, size_t frame_stride, int root, void* hip_stream );

const char* vrc_last_error( void );
/* the kernel instance the calling thread's last vrc_render launched (template arguments spelled as rocprofv3 prints
 * them), "" before the first: lets a benchmark check that a profile it quotes is a profile of what it ran */
const char* vrc_last_kernel( void );
/* ... and how many workgroups of it the runtime says a compute unit holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor:
 * registers, LDS and wave slots together), with the workgroup size: the occupancy the kernels' launch bounds ask for,
 * checkable without a profiler.  The vrc_k_raycast instances only (VRC_EINVAL after another kernel). */
int vrc_last_kernel_occupancy( int* workgroups_per_cu, int* threads_per_workgroup );
/* ABI version of this header */
#define VRC_ABI_VERSION 4 /* 3: vrc_gather_tiles takes the frame height; 4: VRC_KERNEL_PACKED, VRC_OPT_PACKED_ATLAS, vrc_last_kernel_occupancy */
/* = VRC_ABI_VERSION for the product build; -VRC_ABI_VERSION for a developer build of the library (compiled with
 * -DVRC_DEV_BUILD: experiment switches, statistics, ablations that render wrong pixels on purpose) */
int vrc_abi_version( void );
int vrc_is_dev_build( void );

#ifdef __cplusplus
}
#endif
#endif /* VRC_HIP_H */
