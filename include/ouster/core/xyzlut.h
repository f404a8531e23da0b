// xyzlut.h -- XYZ lookup tables and cartesian projection.
//
// Same public surface as the reference ouster_core/include/ouster/core/xyzlut.h
// (make_xyz_lut :53-72, XYZLutT<T> :76-158, cartesian() :173-189) and
// impl/cartesian.h (cartesianT<T> :36-105).  The tables are built once on the host in
// double (the reference does the same, xyzlut.cpp:11-89); the projection itself runs on
// the GPU.  For LUTs built from calibration the device keeps separable per-beam x
// per-column tables instead of the 48 B/pixel LUT; LUTs constructed from user arrays use
// them verbatim.
#pragma once

#include <cstddef>
#include <memory>
#include <stdexcept>
#include <vector>

#include "ouster/core/lidar_frame.h"
#include "ouster/core/typedefs.h"
#include "ouster/core/types.h"

struct ouster_hip_lut;

namespace ouster {
namespace sdk {
namespace core {

template <typename T>
class XYZLutT;
using XYZLut = XYZLutT<double>;

namespace impl {
/** Shared handle to the device-side tables of a LUT. */
struct DeviceLut : std::enable_shared_from_this<DeviceLut> {
    ::ouster_hip_lut* handle = nullptr;
    int device = 0;  ///< GPU the tables live on (ouster::sdk::hip::set_device at creation time)
    ~DeviceLut();
};

/** @throw std::invalid_argument("lut dimensions must be greater than zero") /
 *  ("unexpected frame dimensions") like the reference (xyzlut.cpp:14-21). */
XYZLut make_xyz_lut(size_t w, size_t h, double range_unit, const mat4d& beam_to_lidar_transform,
                    const mat4d& transform, const std::vector<double>& azimuth_angles_deg,
                    const std::vector<double>& altitude_angles_deg);
XYZLut make_xyz_lut(const SensorInfo& sensor, bool use_extrinsics);

/** GPU projection through `dev`: range (h*w u32, staggered) -> points (h*w x 3 of T). */
void cartesian_device(const DeviceLut& dev, const uint32_t* range, size_t n, void* points,
                      bool points_f64);
std::shared_ptr<DeviceLut> device_lut_from_arrays(const void* direction, const void* offset,
                                                  size_t h, size_t w, bool f64);
std::shared_ptr<DeviceLut> device_lut_from_calib(size_t w, size_t h, double range_unit,
                                                 const mat4d& b2l, const mat4d& transform,
                                                 const std::vector<double>& az,
                                                 const std::vector<double>& alt,
                                                 ArrayX3R<double>* direction,
                                                 ArrayX3R<double>* offset);

/** cartesianT (impl/cartesian.h:36-66): points/direction/offset are n x 3 row-major. */
template <typename T>
void cartesianT(ImgRef<T> points, const ImgRef<const uint32_t>& range, const ArrayX3R<T>& direction,
                const ArrayX3R<T>& offset) {
    if (points.rows() != direction.rows() || points.rows() != offset.rows() ||
        points.rows() != range.size())
        throw std::invalid_argument("unexpected image dimensions");
    auto dev = device_lut_from_arrays(direction.data(), offset.data(), range.rows(), range.cols(),
                                      sizeof(T) == 8);
    cartesian_device(*dev, range.data(), range.size(), points.data(), sizeof(T) == 8);
}

template <typename T>
PointCloudXYZ<T> cartesianT(const ImgRef<const uint32_t>& range, const ArrayX3R<T>& direction,
                            const ArrayX3R<T>& offset) {
    if (range.cols() * range.rows() != direction.rows())
        throw std::invalid_argument("unexpected image dimensions");
    PointCloudXYZ<T> points(direction.rows(), impl::uninitialized);
    cartesianT<T>(ImgRef<T>(points), range, direction, offset);
    return points;
}
}  // namespace impl

/** Lookup table of beam directions and offsets (xyzlut.h:76-158). */
template <typename T>
class XYZLutT {
   public:
    const ArrayX3R<T> direction;
    const ArrayX3R<T> offset;
    const size_t h = 0;
    const size_t w = 0;

    template <typename> friend class XYZLutT;

    XYZLutT(const SensorInfo& sensor, bool use_extrinsics = true)
        : XYZLutT(impl::make_xyz_lut(sensor, use_extrinsics)) {}

    /** Converting copy: casts the tables (float LUT = cast of the double LUT, :119-124). */
    template <typename OldT>
    XYZLutT(const XYZLutT<OldT>& o)
        : direction(o.direction.template cast<T>()), offset(o.offset.template cast<T>()), h(o.h),
          w(o.w), dev_(sizeof(T) == sizeof(OldT) ? o.dev_ : nullptr) {}
    XYZLutT(const XYZLutT& o) = default;

    XYZLutT(ArrayX3R<T> direction_, ArrayX3R<T> offset_, size_t h_, size_t w_)
        : direction(std::move(direction_)), offset(std::move(offset_)), h(h_), w(w_) {}

    XYZLutT() = default;

    /** Project a staggered range image; result row i = pixel row*w + col. */
    PointCloudXYZ<T> operator()(const ImgRef<const uint32_t>& range) const {
        if (range.rows() * range.cols() != static_cast<size_t>(direction.rows()))
            throw std::invalid_argument("unexpected image dimensions");
        PointCloudXYZ<T> points(range.rows() * range.cols(), impl::uninitialized);   // pool memory: the kernel writes it in place
        impl::cartesian_device(device(), range.data(), range.size(), points.data(), sizeof(T) == 8);
        return points;
    }
    PointCloudXYZ<T> operator()(const img_t<uint32_t>& range) const {
        return (*this)(ImgRef<const uint32_t>(range));
    }
    PointCloudXYZ<T> operator()(const LidarFrame& frame) const {
        return (*this)(frame.field<uint32_t>(ChanField::RANGE));
    }

    /** Device tables (created on first use; shared between copies). */
    const impl::DeviceLut& device() const {
        if (!dev_) dev_ = impl::device_lut_from_arrays(direction.data(), offset.data(), h, w,
                                                       sizeof(T) == 8);
        return *dev_;
    }
    /** Used by make_xyz_lut to attach the separable device tables. */
    void attach_device(std::shared_ptr<impl::DeviceLut> d) const { dev_ = std::move(d); }

   private:
    mutable std::shared_ptr<impl::DeviceLut> dev_;
};

/** Deprecated free functions of the reference (double only; xyzlut.cpp:111-124). */
PointCloudXYZd cartesian(const LidarFrame& frame, const XYZLut& lut);
PointCloudXYZd cartesian(const ImgRef<const uint32_t>& range, const XYZLut& lut);
inline PointCloudXYZd cartesian(const img_t<uint32_t>& range, const XYZLut& lut) {
    return cartesian(ImgRef<const uint32_t>(range), lut);
}

template <typename T>
std::shared_ptr<const XYZLutT<T>> SensorInfo::xyzlut() const {
    return std::make_shared<const XYZLutT<T>>(XYZLutT<T>(impl::make_xyz_lut(*this, true)));
}

}  // namespace core
}  // namespace sdk
}  // namespace ouster
