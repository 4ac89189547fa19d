/*
 * vrc_comm.hip -- sort-first tile exchange behind the C ABI (include/vrc_hip.h: vrc_comm_*,
 * vrc_gather_tiles): the RGBA32F row bands every rank rendered go to the display rank over RCCL
 * (xGMI inside a node) and land directly at their rows of the frame.
 *
 * Replaces, for a host that does not run Equalizer, what eq::Compositor::assembleFrame does for the
 * reference's 2-D (sort-first) compounds (livre/eq/Channel.cpp:519-523, tiles defined by
 * livre/eq/Channel.cpp:272-290).  No brick data moves between ranks.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1): libvrc_hip.so keeps working on a machine or in
 * a process without it, and a process that already carries an RCCL (PyTorch) shares that instance.
 * xGMI is point-to-point: every peer reaches the display rank over its own link, so the exchange is
 * one ncclGroup of plain sends/receives -- one per band -- rather than a ring collective.
 */
#include "../../include/vrc_hip.h"
#include "vrc_internal.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

static_assert( VRC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "vrc_hip.h: VRC_COMM_ID_BYTES must be RCCL's" );

namespace
{
struct Rccl
{
    void* handle = nullptr;
    ncclResult_t ( *GetUniqueId )( ncclUniqueId* ) = nullptr;
    ncclResult_t ( *CommInitRank )( ncclComm_t*, int, ncclUniqueId, int ) = nullptr;
    ncclResult_t ( *CommDestroy )( ncclComm_t ) = nullptr;
    ncclResult_t ( *GroupStart )() = nullptr;
    ncclResult_t ( *GroupEnd )() = nullptr;
    ncclResult_t ( *Send )( const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    ncclResult_t ( *Recv )( void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    const char* ( *GetErrorString )( ncclResult_t ) = nullptr;
    std::string error;
};

Rccl g_rccl;

Rccl* rccl()
{
    static std::once_flag once;
    Rccl& r = g_rccl;
    std::call_once( once, [&r] {
        /* VRC_RCCL_LIBRARY: another build of RCCL (or the test double of tests/host_san/fake_rccl.cpp) */
        const char* const forced = getenv( "VRC_RCCL_LIBRARY" );
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        if( forced && forced[0] )
            r.handle = dlopen( forced, RTLD_NOW | RTLD_LOCAL );
        else
            for( const char* n : names )
            {
                r.handle = dlopen( n, RTLD_NOW | RTLD_GLOBAL );
                if( r.handle )
                    break;
            }
        if( !r.handle )
        {
            const char* e = dlerror();
            r.error = std::string( "RCCL not found (" ) + ( forced && forced[0] ? forced : "librccl.so.1" ) + "): " + ( e ? e : "?" );
            return;
        }
        bool ok = true;
        auto sym = [&]( const char* name ) {
            void* p = dlsym( r.handle, name );
            if( !p )
            {
                ok = false;
                r.error = std::string( "RCCL lacks " ) + name;
            }
            return p;
        };
        r.GetUniqueId = (decltype( r.GetUniqueId ))sym( "ncclGetUniqueId" );
        r.CommInitRank = (decltype( r.CommInitRank ))sym( "ncclCommInitRank" );
        r.CommDestroy = (decltype( r.CommDestroy ))sym( "ncclCommDestroy" );
        r.GroupStart = (decltype( r.GroupStart ))sym( "ncclGroupStart" );
        r.GroupEnd = (decltype( r.GroupEnd ))sym( "ncclGroupEnd" );
        r.Send = (decltype( r.Send ))sym( "ncclSend" );
        r.Recv = (decltype( r.Recv ))sym( "ncclRecv" );
        r.GetErrorString = (decltype( r.GetErrorString ))sym( "ncclGetErrorString" );
        if( !ok )
        {
            dlclose( r.handle );
            r.handle = nullptr;
        }
    } );
    return r.handle ? &r : nullptr;
}

/* after rccl() returned NULL: why */
std::string rcclWhy() { return g_rccl.error.empty() ? std::string( "librccl.so.1 not loadable" ) : g_rccl.error; }

int rcclFail( const char* what, ncclResult_t e )
{
    Rccl* r = rccl();
    return vrc_internal_fail( VRC_ECOMM, std::string( what ) + ": " +
                                             ( r && r->GetErrorString ? r->GetErrorString( e ) : "RCCL error" ) );
}

#define VRC_RCCL_CHECK( what, expr )          \
    do                                        \
    {                                         \
        const ncclResult_t _e = ( expr );     \
        if( _e != ncclSuccess )               \
            return rcclFail( what, _e );      \
    } while( 0 )
} // namespace

struct vrc_comm
{
    ncclComm_t comm = nullptr; /* NULL for a world of one: nothing to exchange */
    int rank = 0, world = 1, device = 0;
    hipEvent_t ordered = nullptr; /* vrc_gather_tiles on a caller's stream: "the render stream up to here" */
};

extern "C" {

int vrc_comm_unique_id( uint8_t id[VRC_COMM_ID_BYTES] )
{
    if( !id )
        return vrc_internal_fail( VRC_EINVAL, "vrc_comm_unique_id: id is NULL" );
    Rccl* r = rccl();
    if( !r )
        return vrc_internal_fail( VRC_ECOMM, "vrc_comm_unique_id: RCCL unavailable: " + rcclWhy() );
    ncclUniqueId u;
    VRC_RCCL_CHECK( "ncclGetUniqueId", r->GetUniqueId( &u ) );
    ::memcpy( id, u.internal, VRC_COMM_ID_BYTES );
    return VRC_OK;
}

int vrc_comm_create( vrc_ctx* ctx, int rank, int world, const uint8_t id[VRC_COMM_ID_BYTES], vrc_comm** out )
{
    if( !out )
        return vrc_internal_fail( VRC_EINVAL, "vrc_comm_create: out is NULL" );
    *out = nullptr;
    if( !ctx || world < 1 || rank < 0 || rank >= world )
        return vrc_internal_fail( VRC_EINVAL, "vrc_comm_create: bad ctx / rank / world" );
    int device = 0;
    (void)vrc_internal_ctx_stream( ctx, &device );
    vrc_comm* c = new vrc_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    if( world > 1 )
    {
        Rccl* r = rccl();
        if( !r || !id )
        {
            delete c;
            return vrc_internal_fail( VRC_ECOMM, !id ? std::string( "vrc_comm_create: id is NULL" )
                                                     : "vrc_comm_create: RCCL unavailable: " + rcclWhy() );
        }
        const hipError_t he = hipSetDevice( device );
        if( he != hipSuccess )
        {
            delete c;
            return vrc_internal_fail( VRC_EHIP, std::string( "vrc_comm_create: hipSetDevice: " ) + hipGetErrorString( he ) );
        }
        ncclUniqueId u;
        ::memcpy( u.internal, id, VRC_COMM_ID_BYTES );
        const ncclResult_t e = r->CommInitRank( &c->comm, world, u, rank );
        if( e != ncclSuccess )
        {
            delete c;
            return rcclFail( "ncclCommInitRank", e );
        }
    }
    *out = c;
    return VRC_OK;
}

void vrc_comm_destroy( vrc_comm* c )
{
    if( !c )
        return;
    if( c->comm )
    {
        Rccl* r = rccl();
        if( r )
            (void)r->CommDestroy( c->comm );
    }
    if( c->ordered )
        (void)hipEventDestroy( c->ordered );
    delete c;
}

int vrc_comm_info( const vrc_comm* c, int* rank, int* world )
{
    if( !c )
        return vrc_internal_fail( VRC_EINVAL, "vrc_comm_info: comm is NULL" );
    if( rank ) *rank = c->rank;
    if( world ) *world = c->world;
    return VRC_OK;
}

int vrc_gather_tiles( vrc_ctx* ctx, vrc_comm* c, const vrc_band* bands, uint32_t nBands, uint32_t width,
                      uint32_t height, uint32_t nFrames, const void* local, size_t localFrameStride, void* frame,
                      size_t frameStride, int root, void* hipStream )
{
    if( !ctx || !c || ( !bands && nBands ) )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: NULL argument" );
    if( root < 0 || root >= c->world || nFrames == 0 || width == 0 || height == 0 )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: bad root / frame count / width / height" );
    const bool isRoot = c->rank == root;
    if( isRoot && !frame )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: the display rank needs a frame" );
    const size_t rowBytes = (size_t)width * sizeof( vrc_f4 );
    /* validate before anything is queued: a rank that bailed out half-way would leave its peers waiting */
    size_t localRows = 0;
    for( uint32_t b = 0; b < nBands; ++b )
    {
        if( bands[b].rank >= (uint32_t)c->world )
            return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: band of a rank outside the communicator" );
        /* in 64 bits: frame_row = 0xFFFFFFFF with rows = 2 must not wrap into the frame (a device write outside
         * the frame on the display rank otherwise) */
        if( (uint64_t)bands[b].frame_row + (uint64_t)bands[b].rows > (uint64_t)height )
            return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: band outside the frame" );
        if( bands[b].rank == (uint32_t)c->rank )
            localRows += bands[b].rows;
    }
    if( localRows && !local )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: this rank has bands but no local buffer" );
    if( nFrames > 1 && ( localFrameStride < localRows * rowBytes ) )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: local frame stride smaller than a frame's bands" );
    if( isRoot && nFrames > 1 && frameStride < (size_t)height * rowBytes )
        return vrc_internal_fail( VRC_EINVAL, "vrc_gather_tiles: frame stride smaller than a frame" );

    int device = 0;
    const hipStream_t ctxStream = vrc_internal_ctx_stream( ctx, &device );
    hipStream_t stream = hipStream ? (hipStream_t)hipStream : ctxStream;
    const hipError_t he = hipSetDevice( device );
    if( he != hipSuccess )
        return vrc_internal_fail( VRC_EHIP, std::string( "vrc_gather_tiles: hipSetDevice: " ) + hipGetErrorString( he ) );
    /* a caller's stream is ordered behind the render stream that produced `local` (the context's own stream when the
     * renders ran there: vrc_set_stream'ed renders already are on the caller's streams) */
    if( stream != ctxStream )
    {
        if( !c->ordered )
        {
            const hipError_t ee = hipEventCreateWithFlags( &c->ordered, hipEventDisableTiming );
            if( ee != hipSuccess )
                return vrc_internal_fail( VRC_EHIP, std::string( "vrc_gather_tiles: hipEventCreate: " ) + hipGetErrorString( ee ) );
        }
        hipError_t ee = hipEventRecord( c->ordered, ctxStream );
        if( ee == hipSuccess )
            ee = hipStreamWaitEvent( stream, c->ordered, 0 );
        if( ee != hipSuccess )
            return vrc_internal_fail( VRC_EHIP, std::string( "vrc_gather_tiles: ordering behind the render stream: " ) +
                                                    hipGetErrorString( ee ) );
    }

    Rccl* r = c->comm ? rccl() : nullptr;
    if( c->comm && !r )
        return vrc_internal_fail( VRC_ECOMM, "vrc_gather_tiles: RCCL unavailable" );
    /* the display rank's own bands: device-to-device copies on the same stream */
    if( isRoot )
        for( uint32_t f = 0; f < nFrames; ++f )
        {
            size_t off = 0;
            for( uint32_t b = 0; b < nBands; ++b )
            {
                if( bands[b].rank != (uint32_t)c->rank )
                    continue;
                const char* src = (const char*)local + f * localFrameStride + off;
                char* dst = (char*)frame + f * frameStride + (size_t)bands[b].frame_row * rowBytes;
                if( src != dst && bands[b].rows )
                {
                    const hipError_t e = hipMemcpyAsync( dst, src, (size_t)bands[b].rows * rowBytes,
                                                         hipMemcpyDeviceToDevice, stream );
                    if( e != hipSuccess )
                        return vrc_internal_fail( VRC_EHIP, std::string( "vrc_gather_tiles: hipMemcpyAsync: " ) +
                                                                hipGetErrorString( e ) );
                }
                off += (size_t)bands[b].rows * rowBytes;
            }
        }
    if( !c->comm )
        return VRC_OK;

    /* one group: sends and receives of one peer pair match in issue order (band order, frame by frame) */
    VRC_RCCL_CHECK( "ncclGroupStart", r->GroupStart() );
    ncclResult_t first = ncclSuccess;
    for( uint32_t f = 0; f < nFrames && first == ncclSuccess; ++f )
    {
        size_t off = 0;
        for( uint32_t b = 0; b < nBands && first == ncclSuccess; ++b )
        {
            const size_t count = (size_t)bands[b].rows * width * 4u; /* floats */
            if( bands[b].rank == (uint32_t)c->rank )
            {
                if( !isRoot && count )
                    first = r->Send( (const char*)local + f * localFrameStride + off, count, ncclFloat, root,
                                     c->comm, stream );
                off += (size_t)bands[b].rows * rowBytes;
            }
            else if( isRoot && count )
                first = r->Recv( (char*)frame + f * frameStride + (size_t)bands[b].frame_row * rowBytes, count,
                                 ncclFloat, (int)bands[b].rank, c->comm, stream );
        }
    }
    const ncclResult_t end = r->GroupEnd();
    if( first != ncclSuccess )
        return rcclFail( "ncclSend/ncclRecv", first );
    if( end != ncclSuccess )
        return rcclFail( "ncclGroupEnd", end );
    return VRC_OK;
}

} /* extern "C" */
