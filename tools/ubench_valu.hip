// Micro-benchmark: issue cost of the VALU/LDS instructions the raycast loop is made of, on gfx950.
// For each op: 16 independent dependency chains, ITERS iterations, cycles from s_memtime
// (clock64) per wave, with 1, 2, 4 or 8 waves resident per SIMD (grid = 256 CUs x 4 SIMDs x W).
// Prints cycles per wave-instruction as seen by one wave, and per-SIMD throughput cost
// (= that / W).   build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define ITERS 2000

#define BODY16( INSTR )                                                                         \
    asm volatile( INSTR( 0 ) INSTR( 1 ) INSTR( 2 ) INSTR( 3 ) INSTR( 4 ) INSTR( 5 ) INSTR( 6 ) \
                  INSTR( 7 ) INSTR( 8 ) INSTR( 9 ) INSTR( 10 ) INSTR( 11 ) INSTR( 12 )         \
                  INSTR( 13 ) INSTR( 14 ) INSTR( 15 )                                          \
                  : "+v"( r[0] ), "+v"( r[1] ), "+v"( r[2] ), "+v"( r[3] ), "+v"( r[4] ),     \
                    "+v"( r[5] ), "+v"( r[6] ), "+v"( r[7] ), "+v"( r[8] ), "+v"( r[9] ),     \
                    "+v"( r[10] ), "+v"( r[11] ), "+v"( r[12] ), "+v"( r[13] ), "+v"( r[14] ), \
                    "+v"( r[15] )                                                              \
                  : "v"( b ), "v"( c ), "s"( sc ) )

#define I_FMA( n ) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_ADD( n ) "v_add_f32 %" #n ", %" #n ", %16\n"
#define I_CVT( n ) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_LSHR( n ) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_LSHL( n ) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define I_AND( n ) "v_and_b32 %" #n ", %16, %" #n "\n"
#define I_MAD24( n ) "v_mad_u32_u24 %" #n ", %" #n ", %18, %16\n"
#define I_MUL24( n ) "v_mul_u32_u24 %" #n ", %" #n ", %16\n"
#define I_ADD3( n ) "v_add3_u32 %" #n ", %" #n ", %16, %17\n"
#define I_LSHLADD( n ) "v_lshl_add_u32 %" #n ", %" #n ", 3, %16\n"
#define I_ADDU( n ) "v_add_u32 %" #n ", %" #n ", %16\n"
#define I_MULLO( n ) "v_mul_lo_u32 %" #n ", %" #n ", %16\n"
#define I_CNDMASK( n ) "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
#define I_MOV( n ) "v_mov_b32 %" #n ", %16\n"
#define I_MAX( n ) "v_max_f32 %" #n ", %" #n ", %16\n"

template < int OP >
__global__ void ub( unsigned long long* out )
{
    float r[16];
    for( int i = 0; i < 16; ++i )
        r[i] = (float)( threadIdx.x + i ) * 0.001f;
    float b = 1.0001f, c = 0.0001f;
    unsigned sc = 17;
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
        if( OP == 0 ) BODY16( I_FMA );
        if( OP == 1 ) BODY16( I_ADD );
        if( OP == 2 ) BODY16( I_CVT );
        if( OP == 3 ) BODY16( I_LSHR );
        if( OP == 4 ) BODY16( I_LSHL );
        if( OP == 5 ) BODY16( I_AND );
        if( OP == 6 ) BODY16( I_MAD24 );
        if( OP == 7 ) BODY16( I_MUL24 );
        if( OP == 8 ) BODY16( I_ADD3 );
        if( OP == 9 ) BODY16( I_LSHLADD );
        if( OP == 10 ) BODY16( I_ADDU );
        if( OP == 11 ) BODY16( I_MULLO );
        if( OP == 12 ) BODY16( I_CNDMASK );
        if( OP == 13 ) BODY16( I_MOV );
        if( OP == 14 ) BODY16( I_MAX );
    }
    const unsigned long long t1 = clock64();
    float acc = 0;
    for( int i = 0; i < 16; ++i ) acc += r[i];
    if( acc == 12345.678f ) out[0] = 1; // keep live
    if( threadIdx.x == 0 )
        out[1 + blockIdx.x] = t1 - t0;
}

// packed-f32 ops need register pairs: separate kernel
#define BODY8P( INSTR )                                                                         \
    asm volatile( INSTR( 0 ) INSTR( 1 ) INSTR( 2 ) INSTR( 3 ) INSTR( 4 ) INSTR( 5 ) INSTR( 6 ) \
                  INSTR( 7 )                                                                   \
                  : "+v"( p[0] ), "+v"( p[1] ), "+v"( p[2] ), "+v"( p[3] ), "+v"( p[4] ),     \
                    "+v"( p[5] ), "+v"( p[6] ), "+v"( p[7] )                                   \
                  : "v"( pb ), "v"( pc ) )
#define I_PKFMA( n ) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD( n ) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_PKMUL( n ) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
typedef float f2 __attribute__( ( ext_vector_type( 2 ) ) );
template < int OP >
__global__ void ubp( unsigned long long* out )
{
    f2 p[8];
    for( int i = 0; i < 8; ++i ) p[i] = f2{ (float)threadIdx.x * 0.001f, (float)i };
    f2 pb = { 1.0001f, 0.9999f }, pc = { 0.0001f, 0.0002f };
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
        if( OP == 0 ) { BODY8P( I_PKFMA ); BODY8P( I_PKFMA ); }
        if( OP == 1 ) { BODY8P( I_PKADD ); BODY8P( I_PKADD ); }
        if( OP == 2 ) { BODY8P( I_PKMUL ); BODY8P( I_PKMUL ); }
    }
    const unsigned long long t1 = clock64();
    float acc = 0;
    for( int i = 0; i < 8; ++i ) acc += p[i].x + p[i].y;
    if( acc == 12345.678f ) out[0] = 1;
    if( threadIdx.x == 0 ) out[1 + blockIdx.x] = t1 - t0;
}

// LDS: ds_read_b128 all lanes same address (broadcast) / distinct 16-entry spread, ds_read_u8
template < int OP >
__global__ void ubl( unsigned long long* out )
{
    __shared__ float4 lut[260];
    for( int i = threadIdx.x; i < 260; i += blockDim.x ) lut[i] = float4{ (float)i, 1, 2, 3 };
    __syncthreads();
    unsigned idx = OP == 0 ? 64u : ( threadIdx.x * 7u ) & 255u;
    float4 acc = { 0, 0, 0, 0 };
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
#pragma unroll
        for( int k = 0; k < 16; ++k )
        {
            const float4 v = lut[( idx + k ) & 255u];
            acc.x += v.x; idx += (unsigned)v.y - 1u;
        }
    }
    const unsigned long long t1 = clock64();
    if( acc.x == 12345.678f ) out[0] = 1;
    if( threadIdx.x == 0 ) out[1 + blockIdx.x] = t1 - t0;
}

template < typename K >
static void run( const char* name, K kernel, int instrPerIter, unsigned long long* dOut )
{
    for( int w : { 1, 2, 4, 8 } )
    {
        const int blocks = 256 * 4 * w; // 64-thread blocks: w waves per SIMD
        hipMemset( dOut, 0, sizeof( unsigned long long ) * ( blocks + 1 ) );
        hipEvent_t e0, e1;
        hipEventCreate( &e0 );
        hipEventCreate( &e1 );
        hipLaunchKernelGGL( kernel, dim3( blocks ), dim3( 64 ), 0, 0, dOut ); /* warm */
        hipEventRecord( e0 );
        hipLaunchKernelGGL( kernel, dim3( blocks ), dim3( 64 ), 0, 0, dOut );
        hipEventRecord( e1 );
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime( &ms, e0, e1 );
        std::vector< unsigned long long > h( blocks + 1 );
        hipMemcpy( h.data(), dOut, sizeof( unsigned long long ) * ( blocks + 1 ), hipMemcpyDeviceToHost );
        std::vector< unsigned long long > v( h.begin() + 1, h.end() );
        std::sort( v.begin(), v.end() );
        const double med = (double)v[v.size() / 2] / ( (double)ITERS * instrPerIter );
        printf( "%-22s waves/SIMD=%d  ticks/instr seen by a wave=%7.2f  per-SIMD cost=%6.2f  kernel %.3f ms -> %.2f ns per SIMD-instr, tick=%.3f ns\n",
                name, w, med, med / w, ms, ms * 1e6 / ( (double)ITERS * instrPerIter * w ),
                ms * 1e6 / (double)v[v.size() / 2] );
    }
}

int main()
{
    unsigned long long* dOut;
    hipMalloc( &dOut, sizeof( unsigned long long ) * ( 256 * 4 * 8 + 1 ) );
    const char* names[] = { "v_fma_f32", "v_add_f32", "v_cvt_i32_f32", "v_lshrrev_b32", "v_lshlrev_b32",
                            "v_and_b32", "v_mad_u32_u24", "v_mul_u32_u24", "v_add3_u32", "v_lshl_add_u32",
                            "v_add_u32", "v_mul_lo_u32", "v_cndmask_b32", "v_mov_b32", "v_max_f32" };
    run( names[0], ub< 0 >, 16, dOut );  run( names[1], ub< 1 >, 16, dOut );
    run( names[2], ub< 2 >, 16, dOut );  run( names[3], ub< 3 >, 16, dOut );
    run( names[4], ub< 4 >, 16, dOut );  run( names[5], ub< 5 >, 16, dOut );
    run( names[6], ub< 6 >, 16, dOut );  run( names[7], ub< 7 >, 16, dOut );
    run( names[8], ub< 8 >, 16, dOut );  run( names[9], ub< 9 >, 16, dOut );
    run( names[10], ub< 10 >, 16, dOut ); run( names[11], ub< 11 >, 16, dOut );
    run( names[12], ub< 12 >, 16, dOut ); run( names[13], ub< 13 >, 16, dOut );
    run( names[14], ub< 14 >, 16, dOut );
    run( "v_pk_fma_f32", ubp< 0 >, 16, dOut );
    run( "v_pk_add_f32", ubp< 1 >, 16, dOut );
    run( "v_pk_mul_f32", ubp< 2 >, 16, dOut );
    run( "ds_read_b128 bcast", ubl< 0 >, 16, dOut );
    run( "ds_read_b128 spread", ubl< 1 >, 16, dOut );
    return 0;
}
