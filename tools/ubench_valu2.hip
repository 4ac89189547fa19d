// Micro-benchmark, round 3: issue cost on gfx950 of the instructions a trilinear sample from LDS is made of
// (byte unpack + convert, weights, lerps, transfer-function lookup, log/exp, selects, SDWA address forms, DPP
// reductions) and of the LDS reads it can be fed by (u8 / u16 at even and odd addresses / b32 / b64 / b128 with
// the lane -> address pattern of an 8x8 pixel tile over a 32-byte-pitch region).  Also checks that LDS reads at
// odd addresses return the right bytes (gfx950 runs compute in unaligned-access mode).
// Same method as tools/ubench_valu.hip: 16 independent chains, ITERS iterations, W waves per SIMD.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_valu2 tools/ubench_valu2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITERS 1000

#define BODY16( INSTR )                                                                                      \
    asm volatile( INSTR( 0 ) INSTR( 1 ) INSTR( 2 ) INSTR( 3 ) INSTR( 4 ) INSTR( 5 ) INSTR( 6 ) INSTR( 7 )    \
                      INSTR( 8 ) INSTR( 9 ) INSTR( 10 ) INSTR( 11 ) INSTR( 12 ) INSTR( 13 ) INSTR( 14 )      \
                          INSTR( 15 )                                                                        \
                  : "+v"( r[0] ), "+v"( r[1] ), "+v"( r[2] ), "+v"( r[3] ), "+v"( r[4] ), "+v"( r[5] ),      \
                    "+v"( r[6] ), "+v"( r[7] ), "+v"( r[8] ), "+v"( r[9] ), "+v"( r[10] ), "+v"( r[11] ),    \
                    "+v"( r[12] ), "+v"( r[13] ), "+v"( r[14] ), "+v"( r[15] )                               \
                  : "v"( b ), "v"( c ), "s"( sc ), "s"( m64 )                                                \
                  : "vcc" )

/* %16 = b (vgpr), %17 = c (vgpr), %18 = sc (sgpr), %19 = m64 (sgpr pair) */
#define SDWA3 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
#define I_FMA( n ) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_FMAC( n ) "v_fmac_f32 %" #n ", %16, %17\n"
#define I_MULF( n ) "v_mul_f32 %" #n ", %" #n ", %16\n"
#define I_SUBF( n ) "v_sub_f32 %" #n ", %" #n ", %16\n"
#define I_UB0( n ) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define I_UB1( n ) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I_UB3( n ) "v_cvt_f32_ubyte3 %" #n ", %" #n "\n"
#define I_CVTFU( n ) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_CVTUF( n ) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_FLOOR( n ) "v_floor_f32 %" #n ", %" #n "\n"
#define I_FRACT( n ) "v_fract_f32 %" #n ", %" #n "\n"
#define I_MED3( n ) "v_med3_f32 %" #n ", %" #n ", %16, %17\n"
#define I_MINF( n ) "v_min_f32 %" #n ", %" #n ", %16\n"
#define I_LOG( n ) "v_log_f32 %" #n ", %" #n "\n"
#define I_EXP( n ) "v_exp_f32 %" #n ", %" #n "\n"
#define I_PERM( n ) "v_perm_b32 %" #n ", %" #n ", %16, %17\n"
#define I_BFE( n ) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define I_ANDOR( n ) "v_and_or_b32 %" #n ", %" #n ", %16, %17\n"
#define I_OR( n ) "v_or_b32 %" #n ", %" #n ", %16\n"
#define I_LSHR( n ) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_LSHL_SDWA( n ) "v_lshlrev_b32_sdwa %" #n ", %16, %" #n SDWA3
#define I_ADD_SDWA( n ) "v_add_u32_sdwa %" #n ", %16, %" #n SDWA3
#define I_MUL24_SDWA( n ) "v_mul_u32_u24_sdwa %" #n ", %16, %" #n SDWA3
#define I_CND_VCC( n ) "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
#define I_CND_S( n ) "v_cndmask_b32_e64 %" #n ", %" #n ", %16, %19\n"
#define I_MAD24( n ) "v_mad_u32_u24 %" #n ", %" #n ", %18, %16\n"
#define I_ADD3( n ) "v_add3_u32 %" #n ", %" #n ", %16, %17\n"
#define I_LSHLADD( n ) "v_lshl_add_u32 %" #n ", %" #n ", 3, %16\n"
#define I_MINU( n ) "v_min_u32 %" #n ", %" #n ", %16\n"
#define I_MOV_DPP( n ) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_MIN_DPP( n ) "v_min_u32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_OR_DPP( n ) "v_or_b32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_OR_BCAST( n ) "v_or_b32_dpp %" #n ", %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_DOT2( n ) "v_dot2_u32_u16 %" #n ", %" #n ", %16, %17\n"
#define I_SUBU( n ) "v_sub_u32 %" #n ", %" #n ", %16\n"
#define I_MAXF( n ) "v_max_f32 %" #n ", %" #n ", %16\n"
#define I_LDEXP( n ) "v_ldexp_f32 %" #n ", %" #n ", %16\n"
#define I_CVTFI( n ) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_ADDF( n ) "v_add_f32 %" #n ", %" #n ", %16\n"
#define I_ADDU( n ) "v_add_u32 %" #n ", %" #n ", %16\n"
#define I_AND( n ) "v_and_b32 %" #n ", %16, %" #n "\n"
#define I_LSHL( n ) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define I_CMP_CND( n ) "v_cmp_gt_f32 vcc, %" #n ", %16\nv_cndmask_b32 %" #n ", %" #n ", %17, vcc\n"
#define I_MADMIX( n ) "v_mad_u32_u16 %" #n ", %" #n ", %16, %17\n"
#define I_XOR( n ) "v_xor_b32 %" #n ", %" #n ", %16\n"
#define I_RCP( n ) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_MULLEG( n ) "v_mul_legacy_f32 %" #n ", %" #n ", %16\n"
#define I_LSHR_SDWA( n ) "v_lshrrev_b32_sdwa %" #n ", %16, %" #n SDWA3
#define I_CVTUB_SDWA( n ) "v_cvt_f32_u32_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"

template < int OP >
__global__ void ub( unsigned long long* out )
{
    float r[16];
    for( int i = 0; i < 16; ++i )
        r[i] = (float)( threadIdx.x + i ) * 0.001f + 1.0f;
    float b = 1.0001f, c = 0.0001f;
    unsigned sc = 17;
    unsigned long long m64 = 0x5555AAAA3333CCCCull;
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
#define CASE( K, M ) if( OP == K ) BODY16( M );
        CASE( 0, I_FMA ) CASE( 1, I_MULF ) CASE( 2, I_SUBF ) CASE( 3, I_UB0 ) CASE( 4, I_UB1 ) CASE( 5, I_UB3 )
        CASE( 6, I_CVTFU ) CASE( 7, I_CVTUF ) CASE( 8, I_FLOOR ) CASE( 9, I_FRACT ) CASE( 10, I_MED3 )
        CASE( 11, I_MINF ) CASE( 12, I_LOG ) CASE( 13, I_EXP ) CASE( 14, I_PERM ) CASE( 15, I_BFE )
        CASE( 16, I_ANDOR ) CASE( 17, I_OR ) CASE( 18, I_LSHR ) CASE( 19, I_LSHL_SDWA ) CASE( 20, I_ADD_SDWA )
        CASE( 21, I_MUL24_SDWA ) CASE( 22, I_CND_VCC ) CASE( 23, I_CND_S ) CASE( 24, I_MAD24 ) CASE( 25, I_ADD3 )
        CASE( 26, I_LSHLADD ) CASE( 27, I_MINU ) CASE( 28, I_MOV_DPP ) CASE( 29, I_MIN_DPP ) CASE( 30, I_OR_DPP )
        CASE( 31, I_OR_BCAST ) CASE( 32, I_DOT2 ) CASE( 33, I_SUBU ) CASE( 34, I_MAXF ) CASE( 35, I_LDEXP )
        CASE( 36, I_CVTFI ) CASE( 37, I_ADDF ) CASE( 38, I_ADDU ) CASE( 39, I_AND ) CASE( 40, I_LSHL )
        CASE( 41, I_CMP_CND ) CASE( 42, I_MADMIX ) CASE( 43, I_XOR ) CASE( 44, I_RCP ) CASE( 45, I_FMAC )
        CASE( 46, I_LSHR_SDWA ) CASE( 47, I_CVTUB_SDWA )
#undef CASE
    }
    const unsigned long long t1 = clock64();
    float acc = 0;
    for( int i = 0; i < 16; ++i )
        acc += r[i];
    if( acc == 12345.678f )
        out[0] = 1;
    if( threadIdx.x == 0 )
        out[1 + blockIdx.x] = t1 - t0;
}

/* v_readlane_b32 into SGPRs: a kernel of its own (scalar destinations) */
__global__ void ub_readlane( unsigned long long* out )
{
    unsigned v = threadIdx.x * 3u;
    unsigned acc = 0;
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
        unsigned s0, s1, s2, s3, s4, s5, s6, s7;
        asm volatile( "v_readlane_b32 %0, %8, 1\nv_readlane_b32 %1, %8, 2\nv_readlane_b32 %2, %8, 3\n"
                      "v_readlane_b32 %3, %8, 4\nv_readlane_b32 %4, %8, 5\nv_readlane_b32 %5, %8, 6\n"
                      "v_readlane_b32 %6, %8, 7\nv_readlane_b32 %7, %8, 8\n"
                      "v_readlane_b32 %0, %8, 11\nv_readlane_b32 %1, %8, 12\nv_readlane_b32 %2, %8, 13\n"
                      "v_readlane_b32 %3, %8, 14\nv_readlane_b32 %4, %8, 15\nv_readlane_b32 %5, %8, 16\n"
                      "v_readlane_b32 %6, %8, 17\nv_readlane_b32 %7, %8, 18\n"
                      : "=s"( s0 ), "=s"( s1 ), "=s"( s2 ), "=s"( s3 ), "=s"( s4 ), "=s"( s5 ), "=s"( s6 ), "=s"( s7 )
                      : "v"( v ) );
        acc += s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    }
    const unsigned long long t1 = clock64();
    if( acc == 0x12345u )
        out[0] = 1;
    if( threadIdx.x == 0 )
        out[1 + blockIdx.x] = t1 - t0;
}

/* LDS reads: lane -> address pattern of an 8x8 pixel tile (Morton lanes) at a pixel pitch of 1.5 voxels over a
 * region of 32-byte rows; MIS adds an odd byte offset.  16 reads in flight, then one wait. */
#define LDSBODY( INSTR, REGS )                                                                               \
    asm volatile( INSTR( 0, 0 ) INSTR( 1, 64 ) INSTR( 2, 128 ) INSTR( 3, 192 ) INSTR( 4, 768 ) INSTR( 5, 832 ) \
                      INSTR( 6, 896 ) INSTR( 7, 960 ) INSTR( 8, 1536 ) INSTR( 9, 1600 ) INSTR( 10, 1664 )    \
                          INSTR( 11, 1728 ) INSTR( 12, 2304 ) INSTR( 13, 2368 ) INSTR( 14, 2432 )            \
                              INSTR( 15, 2496 ) "s_waitcnt lgkmcnt(0)\n"                                     \
                  : REGS                                                                                     \
                  : "v"( addr ) : "memory" )
#define R1 "=v"( q[0] ), "=v"( q[1] ), "=v"( q[2] ), "=v"( q[3] ), "=v"( q[4] ), "=v"( q[5] ), "=v"( q[6] ), \
           "=v"( q[7] ), "=v"( q[8] ), "=v"( q[9] ), "=v"( q[10] ), "=v"( q[11] ), "=v"( q[12] ),            \
           "=v"( q[13] ), "=v"( q[14] ), "=v"( q[15] )
#define L_U8( n, o ) "ds_read_u8 %" #n ", %16 offset:" #o "\n"
#define L_U16( n, o ) "ds_read_u16 %" #n ", %16 offset:" #o "\n"
#define L_U8D16( n, o ) "ds_read_u8_d16 %" #n ", %16 offset:" #o "\n"
#define L_B32( n, o ) "ds_read_b32 %" #n ", %16 offset:" #o "\n"

template < int OP, int PATTERN >
__global__ void ubl( unsigned long long* out )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char lds[16384];
    for( int i = threadIdx.x; i < 16384; i += blockDim.x )
        lds[i] = (unsigned char)( i * 7 + 3 );
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned lx = ( lane & 1u ) | ( ( lane >> 1 ) & 2u ) | ( ( lane >> 2 ) & 4u );
    const unsigned ly = ( ( lane >> 1 ) & 1u ) | ( ( lane >> 2 ) & 2u ) | ( ( lane >> 3 ) & 4u );
    /* PATTERN 0: tile at 1.5 voxels per pixel, even x; 1: the same + 1 (odd addresses for every lane);
     * 2: lane i -> byte 4 i (conflict-free dwords); 3: every lane the same address; 4: tile at 1 voxel per pixel
     * with x offsets 0..7 (every other lane odd) */
    unsigned addr;
    if( PATTERN == 0 ) addr = ( ( ly * 3u ) / 2u ) * 32u + ( ( lx * 3u ) / 2u & ~1u );
    if( PATTERN == 1 ) addr = ( ( ly * 3u ) / 2u ) * 32u + ( ( lx * 3u ) / 2u | 1u );
    if( PATTERN == 2 ) addr = lane * 4u;
    if( PATTERN == 3 ) addr = 100u;
    if( PATTERN == 4 ) addr = ly * 32u + lx + 3u;
    addr += (unsigned)(uintptr_t)lds; /* LDS offset of the array (0 here, keeps the compiler honest) */
    unsigned q[16];
    unsigned acc = 0;
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
        if( OP == 0 ) LDSBODY( L_U8, R1 );
        if( OP == 1 ) LDSBODY( L_U16, R1 );
        if( OP == 2 ) LDSBODY( L_U8D16, R1 );
        if( OP == 3 ) LDSBODY( L_B32, R1 );
        acc += q[0] ^ q[5] ^ q[15];
    }
    const unsigned long long t1 = clock64();
    if( acc == 0x12345u )
        out[0] = 1;
    if( threadIdx.x == 0 )
        out[1 + blockIdx.x] = t1 - t0;
}

/* wide LDS reads of a table by data-dependent index (the transfer function): b64 / b128 per lane */
template < int BYTES, int SPREAD >
__global__ void ubt( unsigned long long* out )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) float tab[260 * 4];
    for( int i = threadIdx.x; i < 260 * 4; i += blockDim.x )
        tab[i] = (float)i;
    __syncthreads();
    unsigned idx = SPREAD ? ( threadIdx.x * 37u ) & 255u : 64u;
    float acc = 0;
    const unsigned long long t0 = clock64();
    for( int it = 0; it < ITERS; ++it )
    {
#pragma unroll
        for( int k = 0; k < 16; ++k )
        {
            const unsigned j = ( idx + k * 5u ) & 255u;
            if( BYTES == 8 )
            {
                const float2 v = *reinterpret_cast< const float2* >( tab + j * 2 );
                acc += v.x;
            }
            else
            {
                const float4 v = *reinterpret_cast< const float4* >( tab + j * 4 );
                acc += v.x;
            }
        }
        idx += (unsigned)acc & 1u;
    }
    const unsigned long long t1 = clock64();
    if( acc == 12345.678f )
        out[0] = 1;
    if( threadIdx.x == 0 )
        out[1 + blockIdx.x] = t1 - t0;
}

/* correctness of misaligned LDS reads */
__global__ void check_unaligned( unsigned* out )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char lds[1024];
    for( int i = threadIdx.x; i < 1024; i += blockDim.x )
        lds[i] = (unsigned char)( i * 7 + 3 );
    __syncthreads();
    const unsigned a = threadIdx.x * 5u + 1u; /* all residues mod 4 */
    unsigned v16, v32;
    unsigned long long v64;
    const unsigned base = (unsigned)(uintptr_t)lds + a;
    asm volatile( "ds_read_u16 %0, %3\nds_read_b32 %1, %3\nds_read_b64 %2, %3\ns_waitcnt lgkmcnt(0)\n"
                  : "=v"( v16 ), "=v"( v32 ), "=v"( v64 )
                  : "v"( base )
                  : "memory" );
    unsigned e16 = 0, e32 = 0;
    unsigned long long e64 = 0;
    for( int k = 0; k < 8; ++k )
    {
        const unsigned long long byte = (unsigned char)( ( a + k ) * 7 + 3 );
        if( k < 2 ) e16 |= (unsigned)byte << ( 8 * k );
        if( k < 4 ) e32 |= (unsigned)byte << ( 8 * k );
        e64 |= byte << ( 8 * k );
    }
    if( v16 != e16 ) atomicAdd( &out[0], 1u );
    if( v32 != e32 ) atomicAdd( &out[1], 1u );
    if( v64 != e64 ) atomicAdd( &out[2], 1u );
}

template < typename K >
static void run( const char* name, K kernel, int instrPerIter, unsigned long long* dOut, int threads = 64 )
{
    for( int w : { 1, 2, 4 } )
    {
        const int wavesPerBlock = threads / 64;
        const int blocks = 256 * 4 * w / wavesPerBlock;
        hipMemset( dOut, 0, sizeof( unsigned long long ) * ( blocks + 1 ) );
        hipEvent_t e0, e1;
        hipEventCreate( &e0 );
        hipEventCreate( &e1 );
        hipLaunchKernelGGL( kernel, dim3( blocks ), dim3( threads ), 0, 0, dOut ); /* warm */
        hipEventRecord( e0 );
        hipLaunchKernelGGL( kernel, dim3( blocks ), dim3( threads ), 0, 0, dOut );
        hipEventRecord( e1 );
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime( &ms, e0, e1 );
        std::vector< unsigned long long > h( blocks + 1 );
        hipMemcpy( h.data(), dOut, sizeof( unsigned long long ) * ( blocks + 1 ), hipMemcpyDeviceToHost );
        std::vector< unsigned long long > v( h.begin() + 1, h.end() );
        std::sort( v.begin(), v.end() );
        printf( "%-34s waves/SIMD=%d  kernel %.3f ms -> %6.2f ns per SIMD-instr (wall), %6.2f per CU-instr\n", name, w,
                ms, ms * 1e6 / ( (double)ITERS * instrPerIter * w ), ms * 1e6 / ( (double)ITERS * instrPerIter * w * 4 ) );
        fflush( stdout );
    }
}

int main( int argc, char** argv )
{
    unsigned long long* dOut;
    hipMalloc( &dOut, sizeof( unsigned long long ) * ( 256 * 4 * 8 + 1 ) );
    {
        unsigned* d;
        hipMalloc( &d, 16 );
        hipMemset( d, 0, 16 );
        hipLaunchKernelGGL( check_unaligned, dim3( 1 ), dim3( 64 ), 0, 0, d );
        unsigned h[4];
        hipMemcpy( h, d, 16, hipMemcpyDeviceToHost );
        printf( "misaligned LDS reads, wrong lanes of 64: ds_read_u16 %u  ds_read_b32 %u  ds_read_b64 %u\n", h[0], h[1], h[2] );
    }
    const bool ldsOnly = argc > 1 && !strcmp( argv[1], "lds" );
    if( !ldsOnly )
    {
#define RUN( K, NAME, N ) run( NAME, ub< K >, N, dOut );
        RUN( 0, "v_fma_f32", 16 ) RUN( 45, "v_fmac_f32", 16 ) RUN( 1, "v_mul_f32", 16 ) RUN( 2, "v_sub_f32", 16 )
        RUN( 37, "v_add_f32", 16 ) RUN( 3, "v_cvt_f32_ubyte0", 16 ) RUN( 4, "v_cvt_f32_ubyte1", 16 )
        RUN( 5, "v_cvt_f32_ubyte3", 16 ) RUN( 6, "v_cvt_f32_u32", 16 ) RUN( 36, "v_cvt_f32_i32", 16 )
        RUN( 7, "v_cvt_u32_f32", 16 ) RUN( 47, "v_cvt_f32_u32_sdwa BYTE_1", 16 )
        RUN( 8, "v_floor_f32", 16 ) RUN( 9, "v_fract_f32", 16 ) RUN( 10, "v_med3_f32", 16 ) RUN( 11, "v_min_f32", 16 )
        RUN( 34, "v_max_f32", 16 ) RUN( 12, "v_log_f32", 16 ) RUN( 13, "v_exp_f32", 16 ) RUN( 44, "v_rcp_f32", 16 )
        RUN( 35, "v_ldexp_f32", 16 )
        RUN( 14, "v_perm_b32", 16 ) RUN( 15, "v_bfe_u32", 16 ) RUN( 16, "v_and_or_b32", 16 ) RUN( 17, "v_or_b32", 16 )
        RUN( 39, "v_and_b32", 16 ) RUN( 43, "v_xor_b32", 16 ) RUN( 18, "v_lshrrev_b32", 16 ) RUN( 40, "v_lshlrev_b32", 16 )
        RUN( 38, "v_add_u32", 16 ) RUN( 33, "v_sub_u32", 16 ) RUN( 27, "v_min_u32", 16 )
        RUN( 19, "v_lshlrev_b32_sdwa BYTE_3", 16 ) RUN( 46, "v_lshrrev_b32_sdwa BYTE_3", 16 )
        RUN( 20, "v_add_u32_sdwa BYTE_3", 16 ) RUN( 21, "v_mul_u32_u24_sdwa BYTE_3", 16 )
        RUN( 24, "v_mad_u32_u24", 16 ) RUN( 42, "v_mad_u32_u16", 16 ) RUN( 25, "v_add3_u32", 16 )
        RUN( 26, "v_lshl_add_u32", 16 ) RUN( 32, "v_dot2_u32_u16", 16 )
        RUN( 22, "v_cndmask_b32 vcc", 16 ) RUN( 23, "v_cndmask_b32_e64 sgpr pair", 16 )
        RUN( 41, "v_cmp_gt_f32 + v_cndmask (pair)", 16 )
        RUN( 28, "v_mov_b32_dpp quad_perm", 16 ) RUN( 29, "v_min_u32_dpp row_shr", 16 )
        RUN( 30, "v_or_b32_dpp row_shr", 16 ) RUN( 31, "v_or_b32_dpp row_bcast15", 16 )
#undef RUN
        run( "v_readlane_b32", ub_readlane, 16, dOut );
    }
    run( "ds_read_u8   tile 1.5 vox/px even", ubl< 0, 0 >, 16, dOut );
    run( "ds_read_u8   tile 1.5 vox/px odd", ubl< 0, 1 >, 16, dOut );
    run( "ds_read_u8   tile 1 vox/px mixed", ubl< 0, 4 >, 16, dOut );
    run( "ds_read_u8   lane*4", ubl< 0, 2 >, 16, dOut );
    run( "ds_read_u8   broadcast", ubl< 0, 3 >, 16, dOut );
    run( "ds_read_u16  tile 1.5 vox/px even", ubl< 1, 0 >, 16, dOut );
    run( "ds_read_u16  tile 1.5 vox/px odd", ubl< 1, 1 >, 16, dOut );
    run( "ds_read_u16  tile 1 vox/px mixed", ubl< 1, 4 >, 16, dOut );
    run( "ds_read_u16  lane*4", ubl< 1, 2 >, 16, dOut );
    run( "ds_read_u8_d16 tile even", ubl< 2, 0 >, 16, dOut );
    run( "ds_read_b32  tile 1.5 vox/px even(al)", ubl< 3, 2 >, 16, dOut );
    run( "ds_read_b32  tile odd (misaligned)", ubl< 3, 1 >, 16, dOut );
    run( "ds_read_b32  tile 1 vox/px mixed", ubl< 3, 4 >, 16, dOut );
    run( "ds_read_b64 table bcast", ubt< 8, 0 >, 16, dOut );
    run( "ds_read_b64 table spread", ubt< 8, 1 >, 16, dOut );
    run( "ds_read_b128 table bcast", ubt< 16, 0 >, 16, dOut );
    run( "ds_read_b128 table spread", ubt< 16, 1 >, 16, dOut );
    /* the same with four waves per workgroup sharing the CU's LDS pipe */
    run( "ds_read_u8  tile even, 256-thr WG", ubl< 0, 0 >, 16, dOut, 256 );
    run( "ds_read_u16 tile odd, 256-thr WG", ubl< 1, 1 >, 16, dOut, 256 );
    return 0;
}
