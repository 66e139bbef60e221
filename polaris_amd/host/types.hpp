// types.hpp -- float32 vector / matrix / quaternion helpers with the arithmetic of the reference's
// types package (types/vector.go, types/matrix.go, types/quaternion.go): every operation is a
// single-precision operation in the reference's order (trigonometry and square roots go through
// double and are rounded once, as Go's float32(math.Sin(float64(x))) does), so matrices built
// here -- instance transforms, their inverses, the camera frustum -- carry the same bits the Go
// host would upload.  Compiled with -ffp-contract=off (Go on amd64 never fuses).
#pragma once

#include <cmath>
#include <cstring>

namespace polaris {
namespace types {

constexpr float floatCmpEpsilon = 1e-10f; // matrix.go:13

struct Vec3 {
	float x = 0, y = 0, z = 0;
	float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
	Vec3 Add(Vec3 b) const { return {x + b.x, y + b.y, z + b.z}; }
	Vec3 Sub(Vec3 b) const { return {x - b.x, y - b.y, z - b.z}; }
	Vec3 Mul(float s) const { return {x * s, y * s, z * s}; }
	float Dot(Vec3 b) const { return x * b.x + y * b.y + z * b.z; }
	Vec3 Cross(Vec3 b) const { return {y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x}; }
	float Len() const { return (float)std::sqrt((double)(x * x + y * y + z * z)); } // vector.go:55-57
	Vec3 Normalize() const { // vector.go:64-70
		const float l = 1.0f / Len();
		if (l < floatCmpEpsilon) return {};
		return {x * l, y * l, z * l};
	}
	float MaxComponent() const { return std::fmax(x, std::fmax(y, z)); }
};
inline Vec3 MinVec3(Vec3 a, Vec3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline Vec3 MaxVec3(Vec3 a, Vec3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }

struct Vec4 { float x = 0, y = 0, z = 0, w = 0; };

struct Mat4 { // column major, matrix.go
	float m[16] = {0};
	static Mat4 Ident() { Mat4 r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
	static Mat4 Scale(Vec3 s) { // Scale4: zero components become 1, matrix.go:42-53
		Mat4 r;
		r.m[0] = s.x == 0 ? 1.0f : s.x; r.m[5] = s.y == 0 ? 1.0f : s.y; r.m[10] = s.z == 0 ? 1.0f : s.z; r.m[15] = 1;
		return r;
	}
	static Mat4 Translate(Vec3 t) { Mat4 r = Ident(); r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; return r; }
	Vec4 Mul4x1(Vec4 v) const { // matrix.go:61-68
		return {m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
		        m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w};
	}
	Mat4 Mul4(const Mat4 &b) const { // matrix.go:71-90
		Mat4 r;
		for (int c = 0; c < 4; c++)
			for (int row = 0; row < 4; row++)
				r.m[4 * c + row] = m[row] * b.m[4 * c] + m[4 + row] * b.m[4 * c + 1] + m[8 + row] * b.m[4 * c + 2] + m[12 + row] * b.m[4 * c + 3];
		return r;
	}
	// Inv, matrix.go:108-138: cofactor expansion in float32; *singular is set when |det| < 1e-10
	// (where the reference silently returns the zero matrix).
	Mat4 Inv(bool *singular = nullptr) const {
		const float *a = m;
		const float det = a[0] * a[5] * a[10] * a[15] - a[0] * a[5] * a[11] * a[14] - a[0] * a[6] * a[9] * a[15] + a[0] * a[6] * a[11] * a[13] +
		                  a[0] * a[7] * a[9] * a[14] - a[0] * a[7] * a[10] * a[13] - a[1] * a[4] * a[10] * a[15] + a[1] * a[4] * a[11] * a[14] +
		                  a[1] * a[6] * a[8] * a[15] - a[1] * a[6] * a[11] * a[12] - a[1] * a[7] * a[8] * a[14] + a[1] * a[7] * a[10] * a[12] +
		                  a[2] * a[4] * a[9] * a[15] - a[2] * a[4] * a[11] * a[13] - a[2] * a[5] * a[8] * a[15] + a[2] * a[5] * a[11] * a[12] +
		                  a[2] * a[7] * a[8] * a[13] - a[2] * a[7] * a[9] * a[12] - a[3] * a[4] * a[9] * a[14] + a[3] * a[4] * a[10] * a[13] +
		                  a[3] * a[5] * a[8] * a[14] - a[3] * a[5] * a[10] * a[12] - a[3] * a[6] * a[8] * a[13] + a[3] * a[6] * a[9] * a[12];
		const float absDet = det < 0 ? -det : det;
		if (singular) *singular = absDet < floatCmpEpsilon;
		if (absDet < floatCmpEpsilon) return Mat4{};
		Mat4 r;
		r.m[0] = -a[7] * a[10] * a[13] + a[6] * a[11] * a[13] + a[7] * a[9] * a[14] - a[5] * a[11] * a[14] - a[6] * a[9] * a[15] + a[5] * a[10] * a[15];
		r.m[1] = a[3] * a[10] * a[13] - a[2] * a[11] * a[13] - a[3] * a[9] * a[14] + a[1] * a[11] * a[14] + a[2] * a[9] * a[15] - a[1] * a[10] * a[15];
		r.m[2] = -a[3] * a[6] * a[13] + a[2] * a[7] * a[13] + a[3] * a[5] * a[14] - a[1] * a[7] * a[14] - a[2] * a[5] * a[15] + a[1] * a[6] * a[15];
		r.m[3] = a[3] * a[6] * a[9] - a[2] * a[7] * a[9] - a[3] * a[5] * a[10] + a[1] * a[7] * a[10] + a[2] * a[5] * a[11] - a[1] * a[6] * a[11];
		r.m[4] = a[7] * a[10] * a[12] - a[6] * a[11] * a[12] - a[7] * a[8] * a[14] + a[4] * a[11] * a[14] + a[6] * a[8] * a[15] - a[4] * a[10] * a[15];
		r.m[5] = -a[3] * a[10] * a[12] + a[2] * a[11] * a[12] + a[3] * a[8] * a[14] - a[0] * a[11] * a[14] - a[2] * a[8] * a[15] + a[0] * a[10] * a[15];
		r.m[6] = a[3] * a[6] * a[12] - a[2] * a[7] * a[12] - a[3] * a[4] * a[14] + a[0] * a[7] * a[14] + a[2] * a[4] * a[15] - a[0] * a[6] * a[15];
		r.m[7] = -a[3] * a[6] * a[8] + a[2] * a[7] * a[8] + a[3] * a[4] * a[10] - a[0] * a[7] * a[10] - a[2] * a[4] * a[11] + a[0] * a[6] * a[11];
		r.m[8] = -a[7] * a[9] * a[12] + a[5] * a[11] * a[12] + a[7] * a[8] * a[13] - a[4] * a[11] * a[13] - a[5] * a[8] * a[15] + a[4] * a[9] * a[15];
		r.m[9] = a[3] * a[9] * a[12] - a[1] * a[11] * a[12] - a[3] * a[8] * a[13] + a[0] * a[11] * a[13] + a[1] * a[8] * a[15] - a[0] * a[9] * a[15];
		r.m[10] = -a[3] * a[5] * a[12] + a[1] * a[7] * a[12] + a[3] * a[4] * a[13] - a[0] * a[7] * a[13] - a[1] * a[4] * a[15] + a[0] * a[5] * a[15];
		r.m[11] = a[3] * a[5] * a[8] - a[1] * a[7] * a[8] - a[3] * a[4] * a[9] + a[0] * a[7] * a[9] + a[1] * a[4] * a[11] - a[0] * a[5] * a[11];
		r.m[12] = a[6] * a[9] * a[12] - a[5] * a[10] * a[12] - a[6] * a[8] * a[13] + a[4] * a[10] * a[13] + a[5] * a[8] * a[14] - a[4] * a[9] * a[14];
		r.m[13] = -a[2] * a[9] * a[12] + a[1] * a[10] * a[12] + a[2] * a[8] * a[13] - a[0] * a[10] * a[13] - a[1] * a[8] * a[14] + a[0] * a[9] * a[14];
		r.m[14] = a[2] * a[5] * a[12] - a[1] * a[6] * a[12] - a[2] * a[4] * a[13] + a[0] * a[6] * a[13] + a[1] * a[4] * a[14] - a[0] * a[5] * a[14];
		r.m[15] = -a[2] * a[5] * a[8] + a[1] * a[6] * a[8] + a[2] * a[4] * a[9] - a[0] * a[6] * a[9] - a[1] * a[4] * a[10] + a[0] * a[5] * a[10];
		const float s = 1 / det;
		for (float &v : r.m) v = v * s;
		return r;
	}
};

// Perspective4, matrix.go:156-161 (fovy is used as given: the reference's degree conversion is
// commented out, so the OBJ camera_fov value is interpreted as radians)
inline Mat4 Perspective4(float fovy, float aspect, float near, float far) {
	const float nmf = near - far, f = (float)(1.0 / std::tan((double)fovy / 2.0));
	Mat4 r;
	r.m[0] = f / aspect; r.m[5] = f; r.m[10] = (near + far) / nmf; r.m[11] = -1;
	r.m[14] = (2.0f * far * near) / nmf;
	return r;
}

inline Mat4 LookAtV(Vec3 eye, Vec3 center, Vec3 up) { // matrix.go:164-178
	const Vec3 f = center.Sub(eye).Normalize(), s = f.Cross(up.Normalize()).Normalize(), u = s.Cross(f);
	Mat4 rot;
	rot.m[0] = s.x; rot.m[1] = u.x; rot.m[2] = -f.x;
	rot.m[4] = s.y; rot.m[5] = u.y; rot.m[6] = -f.y;
	rot.m[8] = s.z; rot.m[9] = u.z; rot.m[10] = -f.z;
	rot.m[15] = 1;
	Mat4 trans = Mat4::Ident();
	trans.m[12] = -eye.x; trans.m[13] = -eye.y; trans.m[14] = -eye.z;
	return rot.Mul4(trans);
}

struct Quat { // quaternion.go
	Vec3 V;
	float W = 1;
	static Quat FromAxisAngle(Vec3 axis, float angle) { // :20-27
		const float s = (float)std::sin((double)(angle * 0.5f)), c = (float)std::cos((double)(angle * 0.5f));
		return {axis.Mul(s), c};
	}
	Vec3 Rotate(Vec3 v) const { // :31-35
		const Vec3 cross = V.Cross(v);
		return v.Add(cross.Mul(2 * W)).Add(V.Mul(2).Cross(cross));
	}
	Quat Mul(Quat q2) const { return {V.Cross(q2.V).Add(q2.V.Mul(W)).Add(V.Mul(q2.W)), W * q2.W - V.Dot(q2.V)}; } // :40-45
	float Len() const { return (float)std::sqrt((double)(W * W + V.x * V.x + V.y * V.y + V.z * V.z)); }
	Quat Normalize() const { // :56-77
		float length = Len();
		float absDelta = 1 - length;
		if (absDelta < 0) absDelta = -absDelta;
		if (absDelta < floatCmpEpsilon) return *this;
		if (length == 0) return Quat{};
		if (std::isinf(length)) length = 3.402823466e+38f;
		return {V.Mul(1 / length), W * 1 / length};
	}
	Mat4 ToMat4() const { // :90-98
		const float w = W, x = V.x, y = V.y, z = V.z;
		Mat4 r;
		r.m[0] = 1 - 2 * y * y - 2 * z * z; r.m[1] = 2 * x * y + 2 * w * z; r.m[2] = 2 * x * z - 2 * w * y;
		r.m[4] = 2 * x * y - 2 * w * z; r.m[5] = 1 - 2 * x * x - 2 * z * z; r.m[6] = 2 * y * z + 2 * w * x;
		r.m[8] = 2 * x * z + 2 * w * y; r.m[9] = 2 * y * z - 2 * w * x; r.m[10] = 1 - 2 * x * x - 2 * y * y;
		r.m[15] = 1;
		return r;
	}
};

} // namespace types
} // namespace polaris
