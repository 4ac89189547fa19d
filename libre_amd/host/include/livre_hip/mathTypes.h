/*
 * mathTypes.h -- the small vector/matrix/box PODs the plugin surface needs, replacing the
 * un-vendored vmmlib types of livre/core/mathTypes.h:35-77.  Matrices are column-major
 * float[16] exactly as vmmlib stores them and the device layer reads them
 * (renderers/cudaRaycaster/cuda/math.cuh:1457-1464).
 */
#ifndef LIVRE_HIP_MATHTYPES_H
#define LIVRE_HIP_MATHTYPES_H

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <ostream>

namespace livre
{
template < typename T, size_t N > struct Vector
{
    T array[N];
    Vector() { for( size_t i = 0; i < N; ++i ) array[i] = T( 0 ); }
    explicit Vector( T v ) { for( size_t i = 0; i < N; ++i ) array[i] = v; }
    Vector( T x, T y ) { static_assert( N == 2, "" ); array[0] = x; array[1] = y; }
    Vector( T x, T y, T z ) { static_assert( N == 3, "" ); array[0] = x; array[1] = y; array[2] = z; }
    Vector( T x, T y, T z, T w ) { static_assert( N == 4, "" ); array[0] = x; array[1] = y; array[2] = z; array[3] = w; }
    template < typename U > Vector( const Vector< U, N >& o ) { for( size_t i = 0; i < N; ++i ) array[i] = T( o.array[i] ); }
    T& operator[]( size_t i ) { return array[i]; }
    const T& operator[]( size_t i ) const { return array[i]; }
    T x() const { return array[0]; }
    T y() const { return array[1]; }
    T z() const { return array[2]; }
    T find_max() const { T m = array[0]; for( size_t i = 1; i < N; ++i ) if( array[i] > m ) m = array[i]; return m; }
    T find_min() const { T m = array[0]; for( size_t i = 1; i < N; ++i ) if( array[i] < m ) m = array[i]; return m; }
    size_t find_max_index() const { size_t k = 0; for( size_t i = 1; i < N; ++i ) if( array[i] > array[k] ) k = i; return k; }
    T product() const { T p = array[0]; for( size_t i = 1; i < N; ++i ) p *= array[i]; return p; }
    T dot( const Vector& o ) const { T s = T( 0 ); for( size_t i = 0; i < N; ++i ) s += array[i] * o.array[i]; return s; }
    T length() const { return T( std::sqrt( double( dot( *this ) ) ) ); }
    bool operator==( const Vector& o ) const { for( size_t i = 0; i < N; ++i ) if( array[i] != o.array[i] ) return false; return true; }
    bool operator!=( const Vector& o ) const { return !( *this == o ); }
    Vector operator+( const Vector& o ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] + o.array[i]; return r; }
    Vector operator-( const Vector& o ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] - o.array[i]; return r; }
    Vector operator*( const Vector& o ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] * o.array[i]; return r; }
    Vector operator/( const Vector& o ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] / o.array[i]; return r; }
    Vector operator+( T s ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] + s; return r; }
    Vector operator-( T s ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] - s; return r; }
    Vector operator*( T s ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] * s; return r; }
    Vector operator/( T s ) const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = array[i] / s; return r; }
    Vector operator-() const { Vector r; for( size_t i = 0; i < N; ++i ) r.array[i] = -array[i]; return r; }
};

typedef Vector< float, 2 > Vector2f;
typedef Vector< float, 3 > Vector3f;
typedef Vector< float, 4 > Vector4f;
typedef Vector< uint32_t, 2 > Vector2ui;
typedef Vector< uint32_t, 3 > Vector3ui;
typedef Vector< uint32_t, 4 > Vector4ui;
typedef Vector< int32_t, 3 > Vector3i;
typedef Vector< int32_t, 4 > Vector4i;
typedef Vector4i PixelViewport; /* livre/core/mathTypes.h:75 */
typedef Vector4f Viewport;      /* livre/core/mathTypes.h:76 */
typedef std::array< float, 2 > Range;

template < typename T, size_t N > std::ostream& operator<<( std::ostream& os, const Vector< T, N >& v )
{
    os << "(";
    for( size_t i = 0; i < N; ++i ) os << ( i ? ", " : "" ) << v.array[i];
    return os << ")";
}

inline Vector3f normalize( const Vector3f& v ) { return v / v.length(); }
inline Vector3f cross( const Vector3f& a, const Vector3f& b )
{
    return Vector3f( a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] );
}

/** Column-major 4x4, vmmlib layout: array[col*4+row]; operator()(row,col). */
struct Matrix4f
{
    float array[16];
    Matrix4f() { for( int i = 0; i < 16; ++i ) array[i] = ( i % 5 == 0 ) ? 1.f : 0.f; }
    Matrix4f( const float* begin, const float* end ) { int i = 0; for( const float* p = begin; p != end && i < 16; ++p ) array[i++] = *p; }
    /** look-at constructor used at livre/core/settings/CameraSettings.cpp:102 */
    Matrix4f( const Vector3f& eye, const Vector3f& lookAt, const Vector3f& up );
    float& operator()( size_t r, size_t c ) { return array[c * 4 + r]; }
    float operator()( size_t r, size_t c ) const { return array[c * 4 + r]; }
    const float* data() const { return array; }
    Matrix4f operator*( const Matrix4f& o ) const;
    Vector4f operator*( const Vector4f& v ) const;
    /** homogeneous transform of a point (w = 1, divided), as vmmlib's Matrix4 * Vector3 */
    Vector3f operator*( const Vector3f& v ) const;
    Matrix4f inverse() const;
    Vector3f getTranslation() const { return Vector3f( array[12], array[13], array[14] ); }
    Vector4f getColumn( size_t c ) const { return Vector4f( array[c * 4], array[c * 4 + 1], array[c * 4 + 2], array[c * 4 + 3] ); }
    void pre_rotate_x( float angle );
    void pre_rotate_y( float angle );
    bool equals( const Matrix4f& o, float tol ) const { for( int i = 0; i < 16; ++i ) if( std::fabs( array[i] - o.array[i] ) > tol ) return false; return true; }
    bool operator==( const Matrix4f& o ) const { return equals( o, 0.f ); }
};

/** glFrustum-style perspective (eq::Frustumf::computePerspectiveMatrix, livre/eq/Channel.cpp:154-155) */
Matrix4f perspectiveFrustum( float l, float r, float b, float t, float n, float f );

template < typename T > struct AABB
{
    Vector< T, 3 > _min, _max;
    AABB() {}
    AABB( const Vector< T, 3 >& mn, const Vector< T, 3 >& mx ) : _min( mn ), _max( mx ) {}
    const Vector< T, 3 >& getMin() const { return _min; }
    const Vector< T, 3 >& getMax() const { return _max; }
    Vector< T, 3 > getSize() const { return _max - _min; }
    Vector< T, 3 > getCenter() const { return ( _min + _max ) * T( 0.5 ); }
};
typedef AABB< float > Boxf;
typedef AABB< uint32_t > Boxui;
typedef AABB< int32_t > Boxi;

/** plane n.x*x + n.y*y + n.z*z + d = 0, as vmml::Vector4f planes of the frustum culler */
struct Plane
{
    float a, b, c, d;
    Plane() : a( 0 ), b( 0 ), c( 0 ), d( 0 ) {}
    Plane( float a_, float b_, float c_, float d_ ) : a( a_ ), b( b_ ), c( c_ ), d( d_ ) {}
    float x() const { return a; }
    float y() const { return b; }
    float z() const { return c; }
    /** point taken as homogeneous (w = 1), as vmmlib's Vector3 -> Vector4 conversion does */
    float dot( const Vector3f& p ) const { return a * p[0] + b * p[1] + c * p[2] + d; }
    float dot( const Vector4f& p ) const { return a * p[0] + b * p[1] + c * p[2] + d * p[3]; }
};
}
#endif
