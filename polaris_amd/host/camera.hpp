// camera.hpp -- C++ restatement of scene.Camera (asset/scene/camera.go): the pinhole camera whose
// Position and four frustum corner rays are the CameraData payload of Tracer.UpdateState
// (tracer/opencl/tracer.go:175-179).  float32 arithmetic in the reference's order (types.hpp).
#pragma once

#include "tracer.hpp"
#include "types.hpp"

namespace polaris {
namespace scene {

enum class CameraDirection { Up, Down, Left, Right, Forward, Backward }; // camera.go:12-19

struct Camera {
	types::Vec3 Position{0, 0, 0}, LookAt{0, 0, -1}, Up{0, 1, 0}; // NewCamera, camera.go:59-67
	float Pitch = 0, Yaw = 0;
	types::Mat4 ViewMat = types::Mat4::Ident(), ProjMat = types::Mat4::Ident();
	types::Vec4 Frustrum[4]; // TL, TR, BL, BR (the reference's spelling)
	float FOV = 45.0f;
	bool InvertY = false;

	Camera(float fov = 45.0f) : FOV(fov) {}

	void SetupProjection(float aspect) { // camera.go:70-73
		ProjMat = types::Perspective4(FOV, aspect, 1, 1000);
		Update();
	}
	void Move(CameraDirection dir, float offset) { // camera.go:76-97
		types::Vec3 delta;
		const types::Vec3 fwd = LookAt.Sub(Position).Normalize();
		switch (dir) {
		case CameraDirection::Up: delta = Up.Mul(offset); break;
		case CameraDirection::Down: delta = Up.Mul(-offset); break;
		case CameraDirection::Left: delta = fwd.Cross(Up).Mul(-offset); break;
		case CameraDirection::Right: delta = fwd.Cross(Up).Mul(offset); break;
		case CameraDirection::Forward: delta = fwd.Mul(offset); break;
		case CameraDirection::Backward: delta = fwd.Mul(-offset); break;
		}
		Position = Position.Add(delta);
		LookAt = LookAt.Add(delta);
		Update();
	}
	void Update() { // camera.go:100-114
		types::Vec3 dir = LookAt.Sub(Position).Normalize();
		const types::Vec3 pitchAxis = dir.Cross(Up);
		const types::Quat pitchQuat = types::Quat::FromAxisAngle(pitchAxis, Pitch), yawQuat = types::Quat::FromAxisAngle(Up, Yaw);
		const types::Quat orient = pitchQuat.Mul(yawQuat).Normalize();
		dir = orient.Rotate(dir);
		LookAt = Position.Add(dir.Mul(1.0f));
		ViewMat = types::LookAtV(Position, LookAt, Up);
		updateFrustrum();
	}
	types::Mat4 InvViewProjMat() const { return ProjMat.Mul4(ViewMat).Inv(); } // camera.go:116-118

	// the CameraData the tracer consumes
	tracer::CameraData Data() const {
		tracer::CameraData d;
		d.eye[0] = Position.x; d.eye[1] = Position.y; d.eye[2] = Position.z;
		for (int i = 0; i < 4; i++) { d.frustum[4 * i] = Frustrum[i].x; d.frustum[4 * i + 1] = Frustrum[i].y; d.frustum[4 * i + 2] = Frustrum[i].z; d.frustum[4 * i + 3] = Frustrum[i].w; }
		return d;
	}

private:
	void updateFrustrum() { // camera.go:123-142
		const types::Mat4 inv = InvViewProjMat();
		const float yUp = InvertY ? -1.0f : 1.0f;
		const float corners[4][2] = {{-1, yUp}, {1, yUp}, {-1, -yUp}, {1, -yUp}};
		for (int i = 0; i < 4; i++) {
			const types::Vec4 v = inv.Mul4x1({corners[i][0], corners[i][1], -1, 1});
			const float s = 1.0f / v.w;
			const types::Vec3 p = types::Vec3{v.x * s, v.y * s, v.z * s}.Sub(Position);
			Frustrum[i] = {p.x, p.y, p.z, 0};
		}
	}
};

} // namespace scene
} // namespace polaris
