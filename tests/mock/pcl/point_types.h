// TEST-ONLY stand-in (tests/mock/README.md): the record layouts of the three PCL point types registration.hpp touches.
#pragma once
namespace pcl {
struct alignas(16) PointXYZ {
    union {
        float data[4];
        struct { float x, y, z; };
    };
    PointXYZ() : data{ 0.f, 0.f, 0.f, 1.f } {}
    PointXYZ(float x_, float y_, float z_) : data{ x_, y_, z_, 1.f } {}
};
struct alignas(16) Normal {
    union {
        float data_n[4];
        float normal[3];
        struct { float normal_x, normal_y, normal_z; };
    };
    union {
        struct { float curvature; };
        float data_c[4];
    };
    Normal() : data_n{ 0.f, 0.f, 0.f, 0.f }, data_c{ 0.f, 0.f, 0.f, 0.f } {}
};
struct FPFHSignature33 {
    float histogram[33];
};
struct PointNormal;
}  // namespace pcl
