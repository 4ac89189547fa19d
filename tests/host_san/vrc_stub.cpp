/* CPU stand-in for the device C ABI (include/vrc_hip.h), ONLY for the ThreadSanitizer run of the
 * host plugin's threading (tests/test_host.py::test_pipeline_threads_under_thread_sanitizer):
 * the slot pool is a mutex-protected free list, "uploads" touch the brick, "renders" do nothing.
 * TEST INFRASTRUCTURE ONLY -- never linked into the product. */
#include <array>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vrc_hip.h"

struct vrc_ctx
{
    int64_t opt[16] = { 0 };
    uint32_t w = 0, h = 0;
};
struct vrc_pool
{
    std::mutex mutex;
    std::vector< std::array< float, 3 > > freeList;
    uint32_t slotDim[3], slots[3];
    size_t slotBytes, atlasBytes;
    unsigned long long checksum = 0;
};
static thread_local std::string g_err;
extern "C" {
const char* vrc_last_error( void ) { return g_err.c_str(); }
int vrc_abi_version( void ) { return VRC_ABI_VERSION; }
int vrc_ctx_create( int, vrc_ctx** out ) { *out = new vrc_ctx(); return VRC_OK; }
void vrc_ctx_destroy( vrc_ctx* c ) { delete c; }
int vrc_ctx_set_stream( vrc_ctx*, void* ) { return VRC_OK; }
int vrc_set_option( vrc_ctx* c, int o, int64_t v ) { if( o < 0 || o > 15 ) return VRC_EINVAL; c->opt[o] = v; return VRC_OK; }
int vrc_get_option( vrc_ctx* c, int o, int64_t* v )
{
    if( o < 0 || o > 15 ) return VRC_EINVAL;
    *v = o == VRC_OPT_KERNEL_USED ? VRC_KERNEL_GRID_DDA : c->opt[o];
    return VRC_OK;
}
int vrc_pool_create( vrc_ctx*, size_t bpv, int, int, size_t, const uint32_t mb[3], size_t maxBytes, vrc_pool** out )
{
    vrc_pool* p = new vrc_pool();
    for( int a = 0; a < 3; ++a ) p->slotDim[a] = ( mb[a] + 7u ) / 8u * 8u;
    p->slotBytes = size_t( p->slotDim[0] ) * p->slotDim[1] * p->slotDim[2] * bpv;
    size_t n = maxBytes / p->slotBytes;
    if( n == 0 ) n = 1;
    p->slots[0] = uint32_t( n ); p->slots[1] = p->slots[2] = 1;
    p->atlasBytes = n * p->slotBytes;
    for( size_t i = n; i-- > 0; )
        p->freeList.push_back( { float( i ) / float( n ), 0.f, 0.f } );
    *out = p;
    return VRC_OK;
}
void vrc_pool_destroy( vrc_pool* p ) { delete p; }
int vrc_pool_copy_to_slot( vrc_pool* p, const void* brick, const uint32_t size[3], float slot[3] )
{
    std::array< float, 3 > s;
    {
        std::lock_guard< std::mutex > lock( p->mutex );
        if( p->freeList.empty() )
        {
            slot[0] = slot[1] = slot[2] = -1.f;
            g_err = "no free slot";
            return VRC_EFULL;
        }
        s = p->freeList.back();
        p->freeList.pop_back();
    }
    const uint8_t* b = static_cast< const uint8_t* >( brick );
    const size_t n = size_t( size[0] ) * size[1] * size[2];
    unsigned long long sum = 0;
    for( size_t i = 0; i < n; i += 97 ) sum += b[i]; /* the borrowed host pointer is read */
    {
        std::lock_guard< std::mutex > lock( p->mutex );
        p->checksum += sum;
    }
    slot[0] = s[0]; slot[1] = s[1]; slot[2] = s[2];
    return VRC_OK;
}
int vrc_pool_copy_to_slot_device( vrc_pool* p, const void* b, const uint32_t size[3], float slot[3] ) { return vrc_pool_copy_to_slot( p, b, size, slot ); }
int vrc_pool_release_slot( vrc_pool* p, const float slot[3] )
{
    std::lock_guard< std::mutex > lock( p->mutex );
    p->freeList.push_back( { slot[0], slot[1], slot[2] } );
    return VRC_OK;
}
int vrc_pool_info( const vrc_pool* p, size_t* sb, uint32_t ad[3], size_t* ab, uint32_t sl[3], uint32_t* fs )
{
    if( sb ) *sb = p->slotBytes;
    if( ab ) *ab = p->atlasBytes;
    for( int a = 0; a < 3; ++a )
    {
        if( ad ) ad[a] = p->slotDim[a] * p->slots[a];
        if( sl ) sl[a] = p->slots[a];
    }
    if( fs )
    {
        std::lock_guard< std::mutex > lock( const_cast< vrc_pool* >( p )->mutex );
        *fs = uint32_t( p->freeList.size() );
    }
    return VRC_OK;
}
int vrc_pool_synchronize( vrc_pool* ) { return VRC_OK; }
int vrc_pool_read_region( vrc_pool*, const uint32_t*, const uint32_t*, void* ) { return VRC_EUNSUPPORTED; }
int vrc_pool_histogram( vrc_pool*, const float*, const uint32_t*, const uint32_t*, uint32_t, uint64_t, uint64_t* ) { return VRC_EUNSUPPORTED; }
int vrc_update( vrc_ctx*, const float*, const float*, uint32_t ) { return VRC_OK; }
int vrc_pre_render( vrc_ctx* c, const vrc_view_data* v ) { c->w = v->glViewport[2]; c->h = v->glViewport[3]; return VRC_OK; }
int vrc_set_framebuffer( vrc_ctx*, void*, uint32_t, uint32_t ) { return VRC_OK; }
int vrc_get_framebuffer( vrc_ctx* c, void** d, uint32_t* w, uint32_t* h ) { if( d ) *d = nullptr; if( w ) *w = c->w; if( h ) *h = c->h; return VRC_OK; }
int vrc_set_row_map( vrc_ctx*, const uint32_t*, uint32_t ) { return VRC_OK; }
int vrc_set_ray_lod( vrc_ctx*, int, float, float ) { return VRC_OK; }
int vrc_render( vrc_ctx*, const vrc_view_data*, const vrc_node_data*, uint32_t, const vrc_render_data*, vrc_pool* ) { return VRC_OK; }
int vrc_post_render( vrc_ctx*, float* ) { return VRC_OK; }
int vrc_synchronize( vrc_ctx* ) { return VRC_OK; }
int vrc_get_stats( vrc_ctx*, vrc_stats* out ) { std::memset( out, 0, sizeof( *out ) ); return VRC_OK; }
int vrc_get_ray_counts( vrc_ctx*, uint32_t counts[8], int* parts ) { std::memset( counts, 0, 32 ); *parts = 0; return VRC_OK; }
}

/* sort-first tile exchange: a world of one rank, nothing to move */
struct vrc_comm
{
    int rank, world;
};
extern "C" {
int vrc_comm_unique_id( uint8_t id[VRC_COMM_ID_BYTES] )
{
    std::memset( id, 7, VRC_COMM_ID_BYTES );
    return VRC_OK;
}
int vrc_comm_create( vrc_ctx*, int rank, int world, const uint8_t*, vrc_comm** out )
{
    if( world != 1 )
        return VRC_ECOMM;
    *out = new vrc_comm{ rank, world };
    return VRC_OK;
}
void vrc_comm_destroy( vrc_comm* c ) { delete c; }
int vrc_comm_info( const vrc_comm* c, int* rank, int* world )
{
    if( rank ) *rank = c->rank;
    if( world ) *world = c->world;
    return VRC_OK;
}
int vrc_gather_tiles( vrc_ctx*, vrc_comm*, const vrc_band*, uint32_t, uint32_t, uint32_t, uint32_t, const void*, size_t, void*,
                      size_t, int, void* )
{
    return VRC_OK;
}
}
