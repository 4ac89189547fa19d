/*
 * fake_rccl.cpp -- test double for the eight RCCL entry points libvrc_hip.so binds (vrc_comm.hip), for ranks that
 * are THREADS of one process on one GPU (tests/gpu_fake_rccl_gather.py).  It checks what the real library would
 * deadlock or corrupt memory on -- every send must meet a receive of the same peer pair, in issue order, with the
 * same element count -- and moves the data with device-to-device copies ordered after the sender's stream.
 * Loaded through VRC_RCCL_LIBRARY; never part of the product.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace
{
struct Op
{
    bool send;
    void* ptr;
    size_t count;
    int peer;
    hipStream_t stream;
};
struct Posted
{
    const void* ptr;
    size_t count;
    hipEvent_t ready;
};
struct FakeComm
{
    int rank, world;
};
std::mutex g_mutex;
std::condition_variable g_cv;
std::map< std::pair< int, int >, std::deque< Posted > > g_mail; /* (from, to) -> sends in issue order */
int g_errors = 0;
thread_local std::vector< std::pair< FakeComm*, Op > > t_group;
thread_local int t_depth = 0;

ncclResult_t flush()
{
    /* sends first (a rank never waits for its own send), then the receives */
    for( auto& e : t_group )
        if( e.second.send )
        {
            hipEvent_t ev;
            if( hipEventCreateWithFlags( &ev, hipEventDisableTiming ) != hipSuccess ||
                hipEventRecord( ev, e.second.stream ) != hipSuccess )
                return ncclUnhandledCudaError;
            std::lock_guard< std::mutex > lock( g_mutex );
            g_mail[{ e.first->rank, e.second.peer }].push_back( { e.second.ptr, e.second.count, ev } );
            g_cv.notify_all();
        }
    for( auto& e : t_group )
        if( !e.second.send )
        {
            Posted p;
            {
                std::unique_lock< std::mutex > lock( g_mutex );
                auto& q = g_mail[{ e.second.peer, e.first->rank }];
                if( !g_cv.wait_for( lock, std::chrono::seconds( 20 ), [&] { return !q.empty(); } ) )
                {
                    std::fprintf( stderr, "fake_rccl: rank %d waits for a send of rank %d that never came\n",
                                  e.first->rank, e.second.peer );
                    ++g_errors;
                    return ncclInternalError;
                }
                p = q.front();
                q.pop_front();
            }
            if( p.count != e.second.count )
            {
                std::fprintf( stderr, "fake_rccl: rank %d receives %zu elements from rank %d, which sent %zu\n",
                              e.first->rank, e.second.count, e.second.peer, p.count );
                ++g_errors;
                return ncclInvalidArgument;
            }
            if( hipStreamWaitEvent( e.second.stream, p.ready, 0 ) != hipSuccess ||
                hipMemcpyAsync( e.second.ptr, p.ptr, p.count * sizeof( float ), hipMemcpyDeviceToDevice,
                                e.second.stream ) != hipSuccess )
                return ncclUnhandledCudaError;
        }
    t_group.clear();
    return ncclSuccess;
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId( ncclUniqueId* id )
{
    std::memset( id, 0, sizeof( *id ) );
    std::memcpy( id->internal, "fake", 4 );
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank( ncclComm_t* comm, int world, ncclUniqueId id, int rank )
{
    if( std::memcmp( id.internal, "fake", 4 ) != 0 || rank < 0 || rank >= world )
        return ncclInvalidArgument;
    *comm = reinterpret_cast< ncclComm_t >( new FakeComm{ rank, world } );
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy( ncclComm_t comm )
{
    delete reinterpret_cast< FakeComm* >( comm );
    return ncclSuccess;
}
ncclResult_t ncclGroupStart()
{
    ++t_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    if( --t_depth > 0 )
        return ncclSuccess;
    return flush();
}
static ncclResult_t post( bool send, void* ptr, size_t count, ncclDataType_t type, int peer, ncclComm_t comm,
                          hipStream_t stream )
{
    FakeComm* c = reinterpret_cast< FakeComm* >( comm );
    if( type != ncclFloat || peer < 0 || peer >= c->world || peer == c->rank || !ptr )
        return ncclInvalidArgument;
    t_group.push_back( { c, Op{ send, ptr, count, peer, stream } } );
    return t_depth > 0 ? ncclSuccess : flush();
}
ncclResult_t ncclSend( const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s )
{
    return post( true, const_cast< void* >( buf ), count, type, peer, comm, s );
}
ncclResult_t ncclRecv( void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t s )
{
    return post( false, buf, count, type, peer, comm, s );
}
const char* ncclGetErrorString( ncclResult_t e ) { return e == ncclSuccess ? "no error" : "fake_rccl error"; }
/* sends nobody received + mismatches seen: 0 after a correct exchange */
int fake_rccl_leftovers()
{
    std::lock_guard< std::mutex > lock( g_mutex );
    int n = g_errors;
    for( auto& kv : g_mail )
        n += (int)kv.second.size();
    return n;
}
}
