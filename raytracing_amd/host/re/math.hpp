// re/math.hpp — value types of the re:: host API (mirror of the reference's public math types,
// Raytracer/Common.h:7-226).  Only what scene construction and Compile() need lives here: per-ray
// evaluation runs on the device (raytracing_amd/csrc/rt_core.hpp).
// Header-only; arithmetic order follows the reference where results feed the device
// (matrix products, GetRotation's float trig, Color(unsigned)).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>

namespace re {

typedef double real;                       // Common.h:7
constexpr real PI = 3.14159265359f;        // Common.h:9 (float literal, widened)

enum class Axis : unsigned int { X = 0, Y = 1, Z = 2 };

struct Vector2 {
    real X = 0, Y = 0;
    constexpr Vector2() = default;
    constexpr Vector2(real x, real y) : X(x), Y(y) {}
};

struct Vector4;
struct Vector3 {
    real X = 0, Y = 0, Z = 0;
    constexpr Vector3() = default;
    constexpr Vector3(real x, real y, real z) : X(x), Y(y), Z(z) {}
    inline Vector3(const Vector4 &v);

    real &operator[](unsigned i) { return i == 0 ? X : (i == 1 ? Y : Z); }
    real operator[](unsigned i) const { return i == 0 ? X : (i == 1 ? Y : Z); }

    real SquaredLength() const { return X * X + Y * Y + Z * Z; }
    real Length() const { return std::sqrt(X * X + Y * Y + Z * Z); }
    Vector3 Normalized() const {  // Common.cpp:47-58: zero vector stays zero
        real l = Length();
        return l > 0 ? Vector3(X / l, Y / l, Z / l) : Vector3();
    }
    Vector3 operator-() const { return {-X, -Y, -Z}; }

    static const Vector3 Forward, Up, Right, Zero, One;
};
inline const Vector3 Vector3::Forward{0, 0, 1};
inline const Vector3 Vector3::Up{0, 1, 0};
inline const Vector3 Vector3::Right{1, 0, 0};
inline const Vector3 Vector3::Zero{0, 0, 0};
inline const Vector3 Vector3::One{1, 1, 1};

inline Vector3 operator+(const Vector3 &a, const Vector3 &b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
inline Vector3 operator-(const Vector3 &a, const Vector3 &b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
inline Vector3 operator*(const Vector3 &a, const Vector3 &b) { return {a.X * b.X, a.Y * b.Y, a.Z * b.Z}; }
inline Vector3 operator/(const Vector3 &a, const Vector3 &b) { return {a.X / b.X, a.Y / b.Y, a.Z / b.Z}; }
inline Vector3 operator*(const Vector3 &a, real t) { return {a.X * t, a.Y * t, a.Z * t}; }
inline Vector3 operator/(const Vector3 &a, real t) { return {a.X / t, a.Y / t, a.Z / t}; }
inline real operator^(const Vector3 &a, const Vector3 &b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }  // dot
inline Vector3 Cross(const Vector3 &b, const Vector3 &c) {
    return {b.Y * c.Z - b.Z * c.Y, b.Z * c.X - b.X * c.Z, b.X * c.Y - b.Y * c.X};
}

struct Vector4 {
    real X = 0, Y = 0, Z = 0, W = 0;
    Vector4() = default;
    Vector4(real x, real y, real z, real w) : X(x), Y(y), Z(z), W(w) {}
    Vector4(const Vector3 &v, real w) : X(v.X), Y(v.Y), Z(v.Z), W(w) {}
};
inline Vector3::Vector3(const Vector4 &v) : X(v.X), Y(v.Y), Z(v.Z) {}

// Row-major 4x4, element (i,j) at [i*4+j] (Common.h:126-141)
struct Matrix4 {
    real Elements[16];
    real &operator[](int i) { return Elements[i]; }
    real operator[](int i) const { return Elements[i]; }

    static Matrix4 Filled(real diag) {
        Matrix4 m;
        for (int i = 0; i < 16; i++) m.Elements[i] = (i % 5 == 0) ? diag : 0.0;
        return m;
    }
    Matrix4 Transpose() const {
        Matrix4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r[i * 4 + j] = Elements[j * 4 + i];
        return r;
    }
    static Matrix4 GetTranslation(const Vector3 &t) {
        Matrix4 r = Filled(1);
        r[3] = t.X; r[7] = t.Y; r[11] = t.Z;
        return r;
    }
    static Matrix4 GetScale(const Vector3 &s) {
        Matrix4 r = Filled(1);
        r[0] = s.X; r[5] = s.Y; r[10] = s.Z;
        return r;
    }
    // Common.cpp:401-423: yaw = Z, pitch = Y, roll = X, trig evaluated in FLOAT (cosf/sinf)
    static Matrix4 GetRotation(const Vector3 &rot) {
        Matrix4 r = Filled(1);
        real cy = ::cosf((float)rot.Z), sy = ::sinf((float)rot.Z);
        real cp = ::cosf((float)rot.Y), sp = ::sinf((float)rot.Y);
        real cr = ::cosf((float)rot.X), sr = ::sinf((float)rot.X);
        r[0] = cy * cp;  r[1] = cy * sp * sr - sy * cr;  r[2] = cy * sp * cr + sy * sr;
        r[4] = sy * cp;  r[5] = sy * sp * sr + cy * cr;  r[6] = sy * sp * cr - cy * sr;
        r[8] = -sp;      r[9] = cp * sr;                 r[10] = cp * cr;
        return r;
    }
    static const Matrix4 Zero, Identity;
};
inline const Matrix4 Matrix4::Zero = Matrix4::Filled(0);
inline const Matrix4 Matrix4::Identity = Matrix4::Filled(1);

// Common.cpp:152-167: accumulate over k in order, starting from 0
inline Matrix4 operator*(const Matrix4 &l, const Matrix4 &r) {
    Matrix4 o = Matrix4::Filled(0);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) o[i * 4 + j] += l[i * 4 + k] * r[k * 4 + j];
    return o;
}

// Colour with channels clamped to [0,1] by the 3-argument constructor (Common.cpp:260-265).
struct Color {
    real R = 0, G = 0, B = 0;
    Color() = default;
    Color(real r, real g, real b) : R(Clamp01(r)), G(Clamp01(g)), B(Clamp01(b)) {}
    Color(const Vector3 &v) : Color(v.X, v.Y, v.Z) {}
    // 0xRRGGBB; the scale factor is a FLOAT quotient widened to real (Common.cpp:267-273)
    Color(unsigned int rgb) {
        const real f = 1.0f / 255.0f;
        B = (rgb & 0xffu) * f;
        G = ((rgb >> 8) & 0xffu) * f;
        R = ((rgb >> 16) & 0xffu) * f;
    }
    static real Clamp01(real v) {  // std::max(0, std::min(1, v)) semantics of Common.h:22-25
        real m = (v < 1.0) ? v : 1.0;
        return (0.0 < m) ? m : 0.0;
    }
    static const Color White, Black;
};
inline const Color Color::White{1.0, 1.0, 1.0};
inline const Color Color::Black{0.0, 0.0, 0.0};

struct Ray {
    Vector3 Origin = Vector3::Zero, Direction = Vector3::Forward;
};

struct BoundingBox {
    Vector3 Min, Max;  // default (0,0,0)/(0,0,0), Common.h:212
    BoundingBox() = default;
    BoundingBox(Vector3 mn, Vector3 mx) : Min(mn), Max(mx) {}
    // Common.cpp:313-338
    void Split(Axis axis, real value, BoundingBox &left, BoundingBox &right) const {
        left = *this;
        right = *this;
        left.Max[(unsigned)axis] = value;
        right.Min[(unsigned)axis] = value;
    }
};

}  // namespace re
