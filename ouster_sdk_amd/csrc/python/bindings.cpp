// bindings.cpp -- pybind11 module `ouster_sdk_amd.core`: the Python face of the C++ mirror
// (SURVEY.md section 8 f-3).  Names and call shapes follow the reference's nanobind module for
// this path (python/src/cpp/client/processing.cpp:340-357 XYZLut, :527-638 destagger,
// :640-790 FrameBatcher; python/src/cpp/client/lidar_frame.cpp LidarFrame; packet.cpp
// PacketFormat) so tests written against `ouster.sdk.core` read the same:
//     lut = core.XYZLut(info); xyz = lut(frame)            # (h, w, 3) float64
//     img = core.destagger(info, frame.field("RANGE"))
//     batch = core.FrameBatcher(info); done = batch(packet, frame)
// std::invalid_argument surfaces as ValueError like in the reference bindings.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>

#include <pybind11/functional.h>

#include "ouster/core/lidar_scan.h"
#include "ouster/hip/frame_stream.h"
#include "ouster/osf/osf.h"
#include "ouster/pcap/indexed_pcap_reader.h"
#include "ouster/pcap/pcap.h"

namespace py = pybind11;
using namespace ouster::sdk::core;

namespace {

py::dtype dtype_of(ChanFieldType t) {
    switch (t) {
        case ChanFieldType::UINT8: return py::dtype::of<uint8_t>();
        case ChanFieldType::UINT16: return py::dtype::of<uint16_t>();
        case ChanFieldType::UINT32: return py::dtype::of<uint32_t>();
        case ChanFieldType::UINT64: return py::dtype::of<uint64_t>();
        case ChanFieldType::INT8: return py::dtype::of<int8_t>();
        case ChanFieldType::INT16: return py::dtype::of<int16_t>();
        case ChanFieldType::INT32: return py::dtype::of<int32_t>();
        case ChanFieldType::INT64: return py::dtype::of<int64_t>();
        case ChanFieldType::FLOAT32: return py::dtype::of<float>();
        case ChanFieldType::FLOAT64: return py::dtype::of<double>();
        case ChanFieldType::FLOAT16: return py::dtype("float16");
        case ChanFieldType::CHAR: return py::dtype("S1");
        default: throw std::invalid_argument("Invalid field for LidarFrame");
    }
}

ChanFieldType tag_of(const py::dtype& d) {
    for (auto t : {ChanFieldType::UINT8, ChanFieldType::UINT16, ChanFieldType::UINT32,
                   ChanFieldType::UINT64, ChanFieldType::INT8, ChanFieldType::INT16,
                   ChanFieldType::INT32, ChanFieldType::INT64, ChanFieldType::FLOAT32,
                   ChanFieldType::FLOAT64, ChanFieldType::FLOAT16})
        if (dtype_of(t).is(d) || dtype_of(t).equal(d)) return t;
    if (d.kind() == 'S') return ChanFieldType::CHAR;
    throw std::invalid_argument("unsupported numpy dtype");
}

// FieldType from a numpy dtype-like: a fixed-width byte string ("S25") is CHAR with its width as one more trailing
// dimension (python/src/cpp/client/client_common.cpp:129-154)
FieldType make_field_type(const std::string& name, const py::object& dt, std::vector<size_t> extra, FieldClass c) {
    const py::dtype d = py::dtype::from_args(dt);
    const ChanFieldType t = tag_of(d);
    if (t == ChanFieldType::CHAR && d.itemsize() > 0) extra.push_back(static_cast<size_t>(d.itemsize()));
    return FieldType(name, t, std::move(extra), c);
}
std::vector<size_t> dims_of(const py::tuple& t) {
    std::vector<size_t> v;
    for (auto h : t) v.push_back(h.cast<size_t>());
    return v;
}
py::tuple tuple_of(const std::vector<size_t>& v) {
    py::tuple t(v.size());
    for (size_t i = 0; i < v.size(); ++i) t[i] = py::int_(v[i]);
    return t;
}

// numpy view over a Field's memory; `owner` keeps the frame alive
py::array field_view(Field& f, py::handle owner) {
    std::vector<py::ssize_t> shape(f.shape().begin(), f.shape().end());
    return py::array(dtype_of(f.tag()), shape, f.get(), owner);
}

mat4d mat_from(const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
    if (a.size() != 16) throw std::invalid_argument("expected a 4x4 matrix");
    return mat4d::FromRowMajor(a.data());
}
py::array_t<double> mat_to(const mat4d& m) {
    py::array_t<double> a({4, 4});
    std::memcpy(a.mutable_data(), m.data(), sizeof(double) * 16);
    return a;
}

const uint8_t* buf_ptr(const py::buffer& b, size_t min_size) {
    py::buffer_info info = b.request();
    if (static_cast<size_t>(info.size * info.itemsize) < min_size)
        throw std::invalid_argument("Incompatible argument: expected a bytearray of size >= " +
                                    std::to_string(min_size));
    return static_cast<const uint8_t*>(info.ptr);
}

// the bytes of a packet argument: a LidarPacket, or anything with the buffer protocol (bytearray, numpy uint8 array)
const uint8_t* packet_bytes(const py::object& o, size_t min_size) {
    if (py::isinstance<LidarPacket>(o)) {
        const LidarPacket& p = o.cast<const LidarPacket&>();
        if (p.buf.size() < min_size)
            throw std::invalid_argument("Incompatible argument: expected a packet of size >= " + std::to_string(min_size));
        return p.buf.data();
    }
    return buf_ptr(o.cast<py::buffer>(), min_size);
}
size_t packet_size(const py::object& o) {
    if (py::isinstance<LidarPacket>(o)) return o.cast<const LidarPacket&>().buf.size();
    const py::buffer_info info = o.cast<py::buffer>().request();
    return static_cast<size_t>(info.size * info.itemsize);
}
uint8_t* packet_bytes_mut(const py::object& o, size_t min_size) {
    if (py::isinstance<LidarPacket>(o)) return const_cast<uint8_t*>(packet_bytes(o, min_size));
    py::buffer_info info = o.cast<py::buffer>().request(true);
    if (static_cast<size_t>(info.size * info.itemsize) < min_size)
        throw std::invalid_argument("Incompatible argument: expected a bytearray of size >= " + std::to_string(min_size));
    return static_cast<uint8_t*>(info.ptr);
}

enum class ColHeaderSel { TIMESTAMP = 0, ENCODER_COUNT = 1, MEASUREMENT_ID = 2, STATUS = 3, FRAME_ID = 4 };

size_t checked_col(const PacketFormat& pf, size_t col) {
    if (col >= static_cast<size_t>(pf.columns_per_packet)) throw std::invalid_argument("col_idx out of bounds");
    return col;
}

template <typename T, typename F>
py::array header_array(const PacketFormat& pf, const uint8_t* pkt, F&& get) {
    py::array_t<T> out(static_cast<py::ssize_t>(pf.columns_per_packet));
    for (int i = 0; i < pf.columns_per_packet; ++i) out.mutable_data()[i] = static_cast<T>(get(pf.nth_col(i, pkt)));
    return std::move(out);
}

// PacketFormat.set_field(packet, name, (H, columns_per_packet) array): python/src/cpp/client/packet.cpp:357-366
template <typename T>
bool try_set_field(const PacketFormat& pf, LidarPacket& p, const std::string& name, const py::array& a) {
    if (!py::dtype::of<T>().equal(a.dtype())) return false;
    auto arr = py::array_t<T, py::array::c_style | py::array::forcecast>::ensure(a);
    if (!arr || arr.ndim() != 2 || arr.shape(0) != pf.pixels_per_column || arr.shape(1) != pf.columns_per_packet)
        throw std::invalid_argument("field dimension mismatch");
    if (p.buf.size() < pf.lidar_packet_size) throw std::invalid_argument("packet smaller than lidar_packet_size");
    // set_block indexes the plane by the first column's measurement id and skips invalid columns: label the columns
    // 0..n-1 and valid for the write, then put their headers back
    const int n = pf.columns_per_packet;
    std::vector<uint16_t> m_ids(n);
    std::vector<uint32_t> statuses(n);
    for (int i = 0; i < n; ++i) {
        uint8_t* col = pf.nth_col(i, p.buf.data());
        m_ids[i] = pf.col_measurement_id(col);
        statuses[i] = pf.col_status(col);
        pf.set_col_measurement_id(col, static_cast<uint16_t>(i));
        pf.set_col_status(col, 0x1);
    }
    pf.set_block<T>(arr.data(), n, name, p.buf.data());
    for (int i = 0; i < n; ++i) {
        uint8_t* col = pf.nth_col(i, p.buf.data());
        pf.set_col_measurement_id(col, m_ids[i]);
        pf.set_col_status(col, statuses[i]);
    }
    return true;
}

// A numpy array over memory of the library's pool (include/ouster_hip.h, ouster_hip_host_alloc): the kernels write the result
// in place (no staging copy, no first-touch faults) and the block goes back to the pool when the array dies.
py::array pool_array(const py::dtype& dt, const std::vector<py::ssize_t>& shape) {
    size_t bytes = static_cast<size_t>(dt.itemsize());
    for (py::ssize_t d : shape) bytes *= static_cast<size_t>(d);
    struct Block { void* p; size_t n; };
    auto* blk = new Block{impl::host_alloc(bytes ? bytes : 1, false), bytes ? bytes : 1};
    if (!blk->p) {
        delete blk;
        throw std::bad_alloc();
    }
    py::capsule owner(blk, [](void* q) {
        auto* b = static_cast<Block*>(q);
        impl::host_free(b->p, b->n);
        delete b;
    });
    return py::array(dt, shape, blk->p, owner);
}

template <typename T>
py::array lut_call(const XYZLutT<T>& lut, const py::object& arg) {
    // the cloud is written straight into the array that is returned (pool memory: one launch, nothing copied)
    py::array out = pool_array(py::dtype::of<T>(), {static_cast<py::ssize_t>(lut.h), static_cast<py::ssize_t>(lut.w), py::ssize_t{3}});
    const uint32_t* range = nullptr;
    py::array_t<uint32_t, py::array::c_style | py::array::forcecast> r;
    if (py::isinstance<LidarFrame>(arg)) {
        const LidarFrame& fr = arg.cast<const LidarFrame&>();
        if (fr.w != lut.w || fr.h != lut.h) throw std::invalid_argument("unexpected image dimensions");
        range = fr.field<uint32_t>(ChanField::RANGE).data();
    } else {
        r = py::array_t<uint32_t, py::array::c_style | py::array::forcecast>::ensure(arg);
        if (!r || r.ndim() != 2) throw std::invalid_argument("Incompatible argument: expected a 2d range image");
        if (static_cast<size_t>(r.shape(0)) != lut.h || static_cast<size_t>(r.shape(1)) != lut.w)
            throw std::invalid_argument("unexpected image dimensions");
        range = r.data();
    }
    if (lut.h * lut.w != static_cast<size_t>(lut.direction.rows())) throw std::invalid_argument("unexpected image dimensions");
    impl::cartesian_device(lut.device(), range, lut.h * lut.w, out.mutable_data(), sizeof(T) == 8);
    return out;
}

template <typename T>
void bind_lut(py::module_& m, const char* name) {
    py::class_<XYZLutT<T>>(m, name)
        .def(py::init([](const SensorInfo& info, bool use_extrinsics) {
                 return XYZLutT<T>(XYZLutT<double>(info, use_extrinsics));
             }),
             py::arg("info"), py::arg("use_extrinsics") = true)
        .def("__call__", &lut_call<T>)
        .def_property_readonly("direction",
                               [](const XYZLutT<T>& l) {
                                   py::array_t<T> a({static_cast<py::ssize_t>(l.direction.rows()), py::ssize_t{3}});
                                   std::memcpy(a.mutable_data(), l.direction.data(), l.direction.size() * sizeof(T));
                                   return a;
                               })
        .def_property_readonly("offset", [](const XYZLutT<T>& l) {
            py::array_t<T> a({static_cast<py::ssize_t>(l.offset.rows()), py::ssize_t{3}});
            std::memcpy(a.mutable_data(), l.offset.data(), l.offset.size() * sizeof(T));
            return a;
        });
}

}  // namespace

PYBIND11_MODULE(core, m) {
    m.doc() = "MI355X implementation of the ouster.sdk.core hot path (decode, destagger, XYZLut)";

    // kept alive past module init: add_custom_profile registers its profile as a new member (python/src/cpp/client/data.cpp:620-643)
    static py::enum_<UDPProfileLidar>* udp_profile_lidar = nullptr;
    udp_profile_lidar = new py::enum_<UDPProfileLidar>(m, "UDPProfileLidar");
    (*udp_profile_lidar)
        .value("LEGACY", UDPProfileLidar::LEGACY)
        .value("RNG19_RFL8_SIG16_NIR16_DUAL", UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_DUAL)
        .value("RNG19_RFL8_SIG16_NIR16", UDPProfileLidar::RNG19_RFL8_SIG16_NIR16)
        .value("RNG15_RFL8_NIR8", UDPProfileLidar::RNG15_RFL8_NIR8)
        .value("FIVE_WORD_PIXEL", UDPProfileLidar::FIVE_WORD_PIXEL)
        .value("FUSA_RNG15_RFL8_NIR8_DUAL", UDPProfileLidar::FUSA_RNG15_RFL8_NIR8_DUAL)
        .value("RNG15_RFL8_NIR8_DUAL", UDPProfileLidar::RNG15_RFL8_NIR8_DUAL)
        .value("RNG15_RFL8_NIR8_ZONE16", UDPProfileLidar::RNG15_RFL8_NIR8_ZONE16)
        .value("RNG19_RFL8_SIG16_NIR16_ZONE16", UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_ZONE16)
        .value("RNG15_RFL8_WIN8", UDPProfileLidar::RNG15_RFL8_WIN8)
        .value("RNG19_RFL8_SIG16_ZONE16_DUAL", UDPProfileLidar::RNG19_RFL8_SIG16_ZONE16_DUAL)
        .value("RNG19_RFL8_SIG16_NIR16_RGB16", UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_RGB16)
        .value("RNG19_RFL8_SIG16_NIR16_RGB16_DUAL", UDPProfileLidar::RNG19_RFL8_SIG16_NIR16_RGB16_DUAL)
        .value("OFF", UDPProfileLidar::OFF)
        .def_static("from_string", [](const std::string& s) { return udp_profile_lidar_of_string(s).value_or(UDPProfileLidar::UNKNOWN); });
    udp_profile_lidar->attr("__str__") = py::cpp_function([](UDPProfileLidar p) { return to_string(p); }, py::is_method(*udp_profile_lidar));
    py::class_<FieldDecodeInfo>(m, "FieldDecodeInfo")
        .def(py::init([](const py::object& dt, size_t offset, uint64_t mask, int shift, int num_elements) {
                 return FieldDecodeInfo{tag_of(py::dtype::from_args(dt)), offset, mask, shift, num_elements};
             }),
             py::arg("dtype_arg"), py::arg("offset"), py::arg("mask"), py::arg("shift"), py::arg("num_elements") = 1)
        .def_property_readonly("ty_tag", [](const FieldDecodeInfo& f) { return dtype_of(f.ty_tag); })
        .def_readwrite("offset", &FieldDecodeInfo::offset)
        .def_readwrite("mask", &FieldDecodeInfo::mask)
        .def_readwrite("shift", &FieldDecodeInfo::shift)
        .def_readwrite("num_elements", &FieldDecodeInfo::num_elements);
    m.def("add_custom_profile",
          [](const std::string& name, const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields, size_t chan_data_size) {
              const UDPProfileLidar nr = add_custom_profile(name, fields, chan_data_size);
              udp_profile_lidar->value(name.c_str(), nr);
              return static_cast<int>(nr);
          },
          py::arg("name"), py::arg("fields"), py::arg("chan_data_size"));
    m.def("add_custom_profile",
          [](int profile_nr, const std::string& name, const std::vector<std::pair<std::string, FieldDecodeInfo>>& fields,
             size_t chan_data_size) {
              add_custom_profile(profile_nr, name, fields, chan_data_size);
              udp_profile_lidar->value(name.c_str(), static_cast<UDPProfileLidar>(profile_nr));
          },
          py::arg("profile_nr"), py::arg("name"), py::arg("fields"), py::arg("chan_data_size"));
    py::enum_<HeaderType>(m, "HeaderType").value("STANDARD", HeaderType::STANDARD).value("FUSA", HeaderType::FUSA);
    py::enum_<UDPProfileIMU>(m, "UDPProfileIMU")
        .value("LEGACY", UDPProfileIMU::LEGACY)
        .value("ACCEL32_GYRO32_NMEA", UDPProfileIMU::ACCEL32_GYRO32_NMEA)
        .value("OFF", UDPProfileIMU::OFF);
    py::enum_<ThermalShutdownStatus>(m, "ThermalShutdownStatus")
        .value("NORMAL", ThermalShutdownStatus::NORMAL)
        .value("IMMINENT", ThermalShutdownStatus::IMMINENT);
    py::enum_<ShotLimitingStatus>(m, "ShotLimitingStatus")
        .value("NORMAL", ShotLimitingStatus::NORMAL)
        .value("IMMINENT", ShotLimitingStatus::IMMINENT)
        .value("REDUCTION_0_10", ShotLimitingStatus::REDUCTION_0_10)
        .value("REDUCTION_10_20", ShotLimitingStatus::REDUCTION_10_20)
        .value("REDUCTION_20_30", ShotLimitingStatus::REDUCTION_20_30)
        .value("REDUCTION_30_40", ShotLimitingStatus::REDUCTION_30_40)
        .value("REDUCTION_40_50", ShotLimitingStatus::REDUCTION_40_50)
        .value("REDUCTION_50_60", ShotLimitingStatus::REDUCTION_50_60)
        .value("REDUCTION_60_70", ShotLimitingStatus::REDUCTION_60_70)
        .value("REDUCTION_70_75", ShotLimitingStatus::REDUCTION_70_75);
    py::enum_<PacketValidationFailure>(m, "PacketValidationFailure")
        .value("NONE", PacketValidationFailure::NONE)
        .value("PACKET_SIZE", PacketValidationFailure::PACKET_SIZE)
        .value("ID", PacketValidationFailure::ID);
    py::enum_<PacketType>(m, "PacketType")
        .value("Unknown", PacketType::Unknown)
        .value("Lidar", PacketType::Lidar)
        .value("Imu", PacketType::Imu)
        .value("Zone", PacketType::Zone);
    // column header selector of PacketFormat.packet_header (python/src/cpp/client/packet.cpp:213-256)
    py::enum_<ColHeaderSel>(m, "ColHeader")
        .value("TIMESTAMP", ColHeaderSel::TIMESTAMP)
        .value("ENCODER_COUNT", ColHeaderSel::ENCODER_COUNT)
        .value("MEASUREMENT_ID", ColHeaderSel::MEASUREMENT_ID)
        .value("STATUS", ColHeaderSel::STATUS)
        .value("FRAME_ID", ColHeaderSel::FRAME_ID);
    py::enum_<FieldClass>(m, "FieldClass")
        .value("PIXEL_FIELD", FieldClass::PIXEL_FIELD)
        .value("COLUMN_FIELD", FieldClass::COLUMN_FIELD)
        .value("PACKET_FIELD", FieldClass::PACKET_FIELD)
        .value("FRAME_FIELD", FieldClass::FRAME_FIELD);
    py::class_<LidarMode>(m, "LidarMode")
        .def(py::init<const std::string&>())
        .def(py::init<unsigned int, unsigned int>())
        .def_readwrite("columns", &LidarMode::columns)
        .def_readwrite("fps", &LidarMode::fps)
        .def("__eq__", [](const LidarMode& a, const LidarMode& b) { return a == b; })
        .def("__str__", [](const LidarMode& a) { return to_string(a); })
        .def_readonly_static("_512x10", &LidarMode::_512x10)
        .def_readonly_static("_512x20", &LidarMode::_512x20)
        .def_readonly_static("_1024x10", &LidarMode::_1024x10)
        .def_readonly_static("_1024x20", &LidarMode::_1024x20)
        .def_readonly_static("_2048x10", &LidarMode::_2048x10)
        .def_readonly_static("_4096x5", &LidarMode::_4096x5);
    py::class_<SensorConfig>(m, "SensorConfig")
        .def(py::init<>())
        .def_readwrite("lidar_mode", &SensorConfig::lidar_mode)
        .def_readwrite("udp_profile_lidar", &SensorConfig::udp_profile_lidar)
        .def_readwrite("udp_profile_imu", &SensorConfig::udp_profile_imu)
        .def_readwrite("udp_port_lidar", &SensorConfig::udp_port_lidar)   // None = not known (a capture reader may guess it)
        .def_readwrite("udp_port_imu", &SensorConfig::udp_port_imu)
        .def_readwrite("udp_port_zm", &SensorConfig::udp_port_zm);
    // lidar_frame.h:36-74; the element type travels as a numpy dtype
    py::class_<FieldType>(m, "FieldType")
        .def(py::init([](const std::string& name, const py::object& dt, const py::tuple& extra, FieldClass c) {
                 return make_field_type(name, dt, dims_of(extra), c);
             }),
             py::arg("name"), py::arg("dtype"), py::arg("extra_dims") = py::tuple(),
             py::arg("field_class") = FieldClass::PIXEL_FIELD)
        .def_readwrite("name", &FieldType::name)
        // python/src/cpp/client/field.cpp:117-138: switching to / from a fixed-width string moves its width in and out of extra_dims
        .def_property("element_type", [](const FieldType& f) { return dtype_of(f.element_type); },
                      [](FieldType& f, const py::object& dt) {
                          const py::dtype d = py::dtype::from_args(dt);
                          if (f.element_type == ChanFieldType::CHAR && !f.extra_dims.empty()) f.extra_dims.pop_back();
                          const ChanFieldType t = tag_of(d);
                          if (t == ChanFieldType::CHAR && d.itemsize() > 0) f.extra_dims.push_back(static_cast<size_t>(d.itemsize()));
                          f.element_type = t;
                      })
        .def_property("extra_dims", [](const FieldType& f) { return tuple_of(f.extra_dims); },
                      [](FieldType& f, const py::tuple& t) { f.extra_dims = dims_of(t); })
        .def_readwrite("field_class", &FieldType::field_class)
        .def("__eq__", [](const FieldType& a, const py::object& b) { return py::isinstance<FieldType>(b) && a == b.cast<const FieldType&>(); })
        .def("__lt__", [](const FieldType& a, const FieldType& b) { return a < b; })
        .def("__str__", [](const FieldType& f) { return to_string(f); })
        .def("__repr__", [](const FieldType& f) { return "<ouster.sdk.client.FieldType " + to_string(f) + ">"; });

    py::class_<DataFormat>(m, "DataFormat")
        .def(py::init<>())
        .def_readwrite("pixels_per_column", &DataFormat::pixels_per_column)
        .def_readwrite("columns_per_packet", &DataFormat::columns_per_packet)
        .def_readwrite("columns_per_frame", &DataFormat::columns_per_frame)
        .def_readwrite("pixel_shift_by_row", &DataFormat::pixel_shift_by_row)
        .def_readwrite("column_window", &DataFormat::column_window)
        .def_readwrite("udp_profile_lidar", &DataFormat::udp_profile_lidar)
        .def_readwrite("header_type", &DataFormat::header_type)
        .def_readwrite("fps", &DataFormat::fps)
        .def_readwrite("udp_profile_imu", &DataFormat::udp_profile_imu)
        .def_readwrite("imu_packets_per_frame", &DataFormat::imu_packets_per_frame)
        .def_readwrite("imu_measurements_per_packet", &DataFormat::imu_measurements_per_packet)
        .def_readwrite("zone_monitoring_enabled", &DataFormat::zone_monitoring_enabled)
        .def("valid_columns_per_frame", &DataFormat::valid_columns_per_frame)
        .def("lidar_packets_per_frame", &DataFormat::lidar_packets_per_frame);

    py::class_<SensorInfo, std::shared_ptr<SensorInfo>>(m, "SensorInfo")
        .def(py::init<>())
        .def(py::init<const std::string&>(), py::arg("metadata_json"))   // sensor_info.h:229
        .def("__eq__", [](const SensorInfo& a, const py::object& b) { return py::isinstance<SensorInfo>(b) && a == b.cast<const SensorInfo&>(); })
        .def("__copy__", [](const SensorInfo& s) { return SensorInfo(s); })
        .def("__deepcopy__", [](const SensorInfo& s, const py::dict&) { return SensorInfo(s); })
        .def_readwrite("config", &SensorInfo::config)
        .def_readwrite("image_rev", &SensorInfo::image_rev)
        .def_readwrite("sn", &SensorInfo::sn)
        .def_readwrite("fw_rev", &SensorInfo::fw_rev)
        .def_readwrite("prod_line", &SensorInfo::prod_line)
        .def_readwrite("format", &SensorInfo::format)
        .def_readwrite("beam_azimuth_angles", &SensorInfo::beam_azimuth_angles)
        .def_readwrite("beam_altitude_angles", &SensorInfo::beam_altitude_angles)
        .def_readwrite("init_id", &SensorInfo::init_id)
        .def_readwrite("lidar_origin_to_beam_origin_mm", &SensorInfo::lidar_origin_to_beam_origin_mm)
        .def_property("beam_to_lidar_transform", [](const SensorInfo& s) { return mat_to(s.beam_to_lidar_transform); },
                      [](SensorInfo& s, const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
                          s.beam_to_lidar_transform = mat_from(a);
                      })
        .def_property("lidar_to_sensor_transform", [](const SensorInfo& s) { return mat_to(s.lidar_to_sensor_transform); },
                      [](SensorInfo& s, const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
                          s.lidar_to_sensor_transform = mat_from(a);
                      })
        .def_property("sensor_to_body", [](const SensorInfo& s) { return mat_to(s.sensor_to_body); },
                      [](SensorInfo& s, const py::array_t<double, py::array::c_style | py::array::forcecast>& a) {
                          s.sensor_to_body = mat_from(a);
                      })
        .def_property_readonly("w", &SensorInfo::w)
        .def_property_readonly("h", &SensorInfo::h);

    py::class_<PacketFormat, std::shared_ptr<PacketFormat>>(m, "PacketFormat")
        .def(py::init<const SensorInfo&>())
        .def(py::init<const DataFormat&>())
        // (held by shared_ptr here, so the cached format is handed out as a copy)
        .def_static("from_info", [](const SensorInfo& info) { return std::make_shared<PacketFormat>(get_format(info)); })
        .def_static("from_data_format", [](const DataFormat& f) { return std::make_shared<PacketFormat>(get_format(f)); })
        .def_readonly("lidar_packet_size", &PacketFormat::lidar_packet_size)
        .def_readonly("columns_per_packet", &PacketFormat::columns_per_packet)
        .def_readonly("pixels_per_column", &PacketFormat::pixels_per_column)
        .def_readonly("packet_header_size", &PacketFormat::packet_header_size)
        .def_readonly("col_header_size", &PacketFormat::col_header_size)
        .def_readonly("col_size", &PacketFormat::col_size)
        .def_readonly("packet_footer_size", &PacketFormat::packet_footer_size)
        .def_readonly("udp_profile_lidar", &PacketFormat::udp_profile_lidar)
        .def_property_readonly("fields",
                               [](const PacketFormat& pf) {
                                   std::vector<std::string> names;
                                   for (auto it = pf.begin(); it != pf.end(); ++it) names.push_back(it->first);
                                   return names;
                               })
        .def_readonly("udp_profile_imu", &PacketFormat::udp_profile_imu)
        .def_readonly("header_type", &PacketFormat::header_type)
        .def_readonly("max_frame_id", &PacketFormat::max_frame_id)
        // header getters: a LidarPacket or a buffer (python/src/cpp/client/packet.cpp)
        .def("packet_type", [](const PacketFormat& pf, const py::object& b) { return pf.packet_type(packet_bytes(b, 32)); })
        .def("frame_id", [](const PacketFormat& pf, const py::object& b) { return pf.frame_id(packet_bytes(b, 32)); })
        .def("init_id", [](const PacketFormat& pf, const py::object& b) { return pf.init_id(packet_bytes(b, 32)); })
        .def("prod_sn", [](const PacketFormat& pf, const py::object& b) { return pf.prod_sn(packet_bytes(b, 32)); })
        .def("alert_flags", [](const PacketFormat& pf, const py::object& b) { return pf.alert_flags(packet_bytes(b, 32)); })
        .def("countdown_thermal_shutdown",
             [](const PacketFormat& pf, const py::object& b) { return pf.countdown_thermal_shutdown(packet_bytes(b, 32)); })
        .def("countdown_shot_limiting",
             [](const PacketFormat& pf, const py::object& b) { return pf.countdown_shot_limiting(packet_bytes(b, 32)); })
        .def("thermal_shutdown",
             [](const PacketFormat& pf, const py::object& b) { return pf.thermal_shutdown(packet_bytes(b, 32)); })
        .def("shot_limiting",
             [](const PacketFormat& pf, const py::object& b) { return pf.shot_limiting(packet_bytes(b, 32)); })
        .def("crc", [](const PacketFormat& pf, const py::object& b) -> py::object {
            const auto v = pf.crc(packet_bytes(b, 32), packet_size(b));
            if (!v) return py::none();
            return py::int_(*v);
        })
        .def("calculate_crc", [](const PacketFormat& pf, const py::object& b) {
            return pf.calculate_crc(packet_bytes(b, 32), packet_size(b));
        })
        .def("frame_id_difference", &PacketFormat::frame_id_difference)
        // column headers by column index
        .def("col_status", [](const PacketFormat& pf, const py::object& b, size_t col) {
            return pf.col_status(pf.nth_col(checked_col(pf, col), packet_bytes(b, pf.lidar_packet_size)));
        })
        .def("col_timestamp", [](const PacketFormat& pf, const py::object& b, size_t col) {
            return pf.col_timestamp(pf.nth_col(checked_col(pf, col), packet_bytes(b, pf.lidar_packet_size)));
        })
        .def("col_measurement_id", [](const PacketFormat& pf, const py::object& b, size_t col) {
            return pf.col_measurement_id(pf.nth_col(checked_col(pf, col), packet_bytes(b, pf.lidar_packet_size)));
        })
        // setters (parsing.cpp:1007-1090), as the reference's Python module spells them: (packet, [column,] value)
        .def("set_col_status", [](const PacketFormat& pf, const py::object& b, size_t col, uint32_t v) {
            pf.set_col_status(pf.nth_col(checked_col(pf, col), packet_bytes_mut(b, pf.lidar_packet_size)), v);
        })
        .def("set_col_timestamp", [](const PacketFormat& pf, const py::object& b, size_t col, uint64_t v) {
            pf.set_col_timestamp(pf.nth_col(checked_col(pf, col), packet_bytes_mut(b, pf.lidar_packet_size)), v);
        })
        .def("set_col_measurement_id", [](const PacketFormat& pf, const py::object& b, size_t col, uint16_t v) {
            pf.set_col_measurement_id(pf.nth_col(checked_col(pf, col), packet_bytes_mut(b, pf.lidar_packet_size)), v);
        })
        .def("set_frame_id", [](const PacketFormat& pf, const py::object& b, uint32_t v) { pf.set_frame_id(packet_bytes_mut(b, 32), v); })
        .def("set_init_id", [](const PacketFormat& pf, const py::object& b, uint32_t v) { pf.set_init_id(packet_bytes_mut(b, 32), v); })
        .def("set_prod_sn", [](const PacketFormat& pf, const py::object& b, uint64_t v) { pf.set_prod_sn(packet_bytes_mut(b, 32), v); })
        .def("set_alert_flags", [](const PacketFormat& pf, const py::object& b, uint8_t v) { pf.set_alert_flags(packet_bytes_mut(b, 32), v); })
        .def("set_shutdown", [](const PacketFormat& pf, const py::object& b, uint8_t v) { pf.set_shutdown(packet_bytes_mut(b, 32), v); })
        .def("set_shot_limiting", [](const PacketFormat& pf, const py::object& b, uint8_t v) { pf.set_shot_limiting(packet_bytes_mut(b, 32), v); })
        .def("set_shutdown_countdown",
             [](const PacketFormat& pf, const py::object& b, uint8_t v) { pf.set_shutdown_countdown(packet_bytes_mut(b, 32), v); })
        .def("set_shot_limiting_countdown",
             [](const PacketFormat& pf, const py::object& b, uint8_t v) { pf.set_shot_limiting_countdown(packet_bytes_mut(b, 32), v); })
        .def("field_value_mask", &PacketFormat::field_value_mask)
        .def("field_bitness", &PacketFormat::field_bitness)
        // python/src/cpp/client/packet.cpp:173-210 -- (H, columns_per_packet) array, GPU decode
        .def("set_field", [](const PacketFormat& pf, LidarPacket& p, const std::string& name, const py::array& a) {
            if (try_set_field<uint8_t>(pf, p, name, a) || try_set_field<uint16_t>(pf, p, name, a) ||
                try_set_field<uint32_t>(pf, p, name, a) || try_set_field<uint64_t>(pf, p, name, a) ||
                try_set_field<int8_t>(pf, p, name, a) || try_set_field<int16_t>(pf, p, name, a) ||
                try_set_field<int32_t>(pf, p, name, a) || try_set_field<int64_t>(pf, p, name, a) ||
                try_set_field<float>(pf, p, name, a) || try_set_field<double>(pf, p, name, a))
                return;
            throw std::invalid_argument("set_field: unsupported array dtype");
        })
        .def("packet_header", [](const PacketFormat& pf, const py::object& header, const py::object& b) -> py::array {
            const uint8_t* pkt = packet_bytes(b, pf.lidar_packet_size);
            const py::object as_int = py::reinterpret_steal<py::object>(PyNumber_Long(header.ptr()));
            if (!as_int) throw py::error_already_set();
            switch (static_cast<ColHeaderSel>(as_int.cast<int>())) {
                case ColHeaderSel::TIMESTAMP: return header_array<uint64_t>(pf, pkt, [&](const uint8_t* c) { return pf.col_timestamp(c); });
                case ColHeaderSel::ENCODER_COUNT: return header_array<uint32_t>(pf, pkt, [&](const uint8_t* c) { return pf.col_encoder(c); });
                case ColHeaderSel::MEASUREMENT_ID: return header_array<uint16_t>(pf, pkt, [&](const uint8_t* c) { return pf.col_measurement_id(c); });
                case ColHeaderSel::STATUS: return header_array<uint32_t>(pf, pkt, [&](const uint8_t* c) { return pf.col_status(c); });
                case ColHeaderSel::FRAME_ID: return header_array<uint16_t>(pf, pkt, [&](const uint8_t* c) { return pf.col_frame_id(c); });
            }
            throw py::key_error("Invalid header index for PacketFormat");
        })
        .def("packet_field", [](const PacketFormat& pf, const std::string& name, const py::object& b) {
            const uint8_t* p = packet_bytes(b, pf.lidar_packet_size);
            const FieldDecodeInfo& info = pf.field_decode_info(name);
            std::vector<uint8_t> pkt(p, p + pf.lidar_packet_size);
            pkt.resize(pkt.size() + 8, 0);
            // relabel the columns 0..cpp-1 so the packet lands in an H x cpp plane
            for (uint32_t i = 0; i < pf.columns_per_packet; ++i) {
                uint8_t* col = pf.nth_col(i, pkt.data());
                pf.set_col_measurement_id(col, static_cast<uint16_t>(i));
                pf.set_col_status(col, pf.col_status(col) | 0x01);
            }
            const int cols = static_cast<int>(pf.columns_per_packet);
            py::array out(dtype_of(info.ty_tag), std::vector<py::ssize_t>{pf.pixels_per_column, cols});
            std::memset(out.mutable_data(), 0, static_cast<size_t>(out.nbytes()));
            switch (field_type_size(info.ty_tag)) {
                case 1: pf.block_field<uint8_t, 4>(static_cast<uint8_t*>(out.mutable_data()), cols, name, pkt.data()); break;
                case 2: pf.block_field<uint16_t, 4>(static_cast<uint16_t*>(out.mutable_data()), cols, name, pkt.data()); break;
                case 4: pf.block_field<uint32_t, 4>(static_cast<uint32_t*>(out.mutable_data()), cols, name, pkt.data()); break;
                default: pf.block_field<uint64_t, 4>(static_cast<uint64_t*>(out.mutable_data()), cols, name, pkt.data());
            }
            return out;
        });

    py::class_<LidarPacket>(m, "LidarPacket")
        .def(py::init<int>(), py::arg("size") = 65536)
        .def(py::init([](const PacketFormat& pf) { return LidarPacket(std::make_shared<PacketFormat>(pf)); }), py::arg("format"))
        .def("__copy__", [](const LidarPacket& p) { return LidarPacket(p); })
        .def("__deepcopy__", [](const LidarPacket& p, const py::dict&) { return LidarPacket(p); })
        .def_property_readonly("format", [](const LidarPacket& p) { return p.format; })
        .def("validate", [](const LidarPacket& p, const SensorInfo& info) { return p.validate(info); })
        .def("validate", [](const LidarPacket& p, const SensorInfo& info, const PacketFormat& pf) { return p.validate(info, pf); })
        .def_property_readonly("type", [](const LidarPacket& p) { return p.type(); })
        .def("packet_type", [](const LidarPacket& p) { return p.packet_type(); })
        .def("frame_id", [](const LidarPacket& p) { return p.frame_id(); })
        .def("init_id", [](const LidarPacket& p) { return p.init_id(); })
        .def("prod_sn", [](const LidarPacket& p) { return p.prod_sn(); })
        .def("alert_flags", [](const LidarPacket& p) { return p.alert_flags(); })
        .def("shot_limiting", [](const LidarPacket& p) { return p.shot_limiting(); })
        .def("thermal_shutdown", [](const LidarPacket& p) { return p.thermal_shutdown(); })
        .def_readwrite("host_timestamp", &LidarPacket::host_timestamp)
        .def_property("buf",
                      [](py::object self) {
                          LidarPacket& p = self.cast<LidarPacket&>();
                          return py::array(py::dtype::of<uint8_t>(), {static_cast<py::ssize_t>(p.buf.size())},
                                           p.buf.data(), self);
                      },
                      [](LidarPacket& p, const py::buffer& b) {
                          py::buffer_info i = b.request();
                          const uint8_t* src = static_cast<const uint8_t*>(i.ptr);
                          p.buf.assign(src, src + i.size * i.itemsize);
                      });

    py::class_<LidarFrame>(m, "LidarFrame")
        .def(py::init<>())
        .def(py::init<const SensorInfo&>())
        // a copy, optionally onto another field set: casts, zero-fills or drops fields (lidar_frame.h:262)
        .def(py::init([](const LidarFrame& src) { return LidarFrame(src); }), py::arg("source"))
        .def(py::init([](const LidarFrame& src, const std::vector<FieldType>& fields) { return LidarFrame(src, fields); }),
             py::arg("source"), py::arg("field_types"))
        .def(py::init([](size_t h, size_t w) {
                 PyErr_WarnEx(PyExc_FutureWarning,
                              "LidarFrame(h, w) is deprecated, use LidarFrame(h, w, field_types, columns_per_packet) instead", 1);
                 return LidarFrame(h, w);
             }),
             py::arg("h"), py::arg("w"))
        .def(py::init([](const SensorInfo& info, const std::vector<FieldType>& fields) {
                 return LidarFrame(std::make_shared<SensorInfo>(info), fields);
             }),
             py::arg("info"), py::arg("field_types"))
        .def(py::init<size_t, size_t, const std::vector<FieldType>&, size_t>(), py::arg("h"), py::arg("w"),
             py::arg("field_types"), py::arg("columns_per_packet") = DEFAULT_COLUMNS_PER_PACKET)
        .def("__copy__", [](const LidarFrame& f) { return LidarFrame(f); })
        .def("__deepcopy__", [](const LidarFrame& f, const py::dict&) { return LidarFrame(f); })
        .def_readwrite("shutdown_countdown", &LidarFrame::shutdown_countdown)
        .def_readwrite("shot_limiting_countdown", &LidarFrame::shot_limiting_countdown)
        .def_readwrite("sensor_info", &LidarFrame::sensor_info)
        .def("shot_limiting", &LidarFrame::shot_limiting)
        .def("thermal_shutdown", &LidarFrame::thermal_shutdown)
        .def("complete", [](const LidarFrame& f, const py::object& window) {
                 if (window.is_none()) return f.complete();
                 const auto w = window.cast<std::pair<int, int>>();
                 return f.complete(ColumnWindow{w.first, w.second});
             },
             py::arg("window") = py::none())
        .def_property_readonly("field_types", &LidarFrame::field_types)
        .def("__repr__", [](const LidarFrame& f) { return to_string(f); })
        .def(py::init<size_t, size_t, UDPProfileLidar, size_t>(), py::arg("h"), py::arg("w"),
             py::arg("profile"), py::arg("columns_per_packet") = DEFAULT_COLUMNS_PER_PACKET)
        .def_readonly("w", &LidarFrame::w)
        .def_readonly("h", &LidarFrame::h)
        .def_readwrite("frame_id", &LidarFrame::frame_id)
        .def_readwrite("frame_status", &LidarFrame::frame_status)
        .def("has_field", &LidarFrame::has_field)
        .def("field", [](py::object self, const std::string& name) {
            return field_view(self.cast<LidarFrame&>().field(name), self);
        })
        // python/src/cpp/client/lidar_frame.cpp:273-370: (name, array[, class]) deep-copies the array in;
        // (name, dtype[, extra_dims, class]) and (FieldType) add a zero-filled field
        .def("add_field",
             [](py::object self, const std::string& name, const py::object& data, FieldClass c) {
                 // only a real array is a value; a dtype-like (np.uint8, "u4", np.dtype(...)) belongs to the next overload
                 if (!py::isinstance<py::array>(data)) throw py::reference_cast_error();
                 LidarFrame& f = self.cast<LidarFrame&>();
                 const py::array src = py::array::ensure(data, py::array::c_style);
                 const py::dtype d = src.dtype();
                 std::vector<size_t> shape(src.shape(), src.shape() + src.ndim());
                 const ChanFieldType t = tag_of(d);
                 if (t == ChanFieldType::CHAR && d.itemsize() > 0) shape.push_back(static_cast<size_t>(d.itemsize()));
                 Field& fld = f.add_field(name, FieldDescriptor::array(t, shape), c);
                 if (fld.bytes()) std::memcpy(fld.get(), src.data(), fld.bytes());
                 return field_view(fld, self);
             },
             py::arg("name"), py::arg("data"), py::arg("field_class") = FieldClass::PIXEL_FIELD)
        .def("add_field",
             [](py::object self, const std::string& name, const py::object& dt, const py::tuple& extra, FieldClass c) {
                 LidarFrame& f = self.cast<LidarFrame&>();
                 return field_view(f.add_field(make_field_type(name, dt, dims_of(extra), c)), self);
             },
             py::arg("name"), py::arg("dtype"), py::arg("shape") = py::tuple(), py::arg("field_class") = FieldClass::PIXEL_FIELD)
        .def("add_field", [](py::object self, const FieldType& t) { return field_view(self.cast<LidarFrame&>().add_field(t), self); },
             py::arg("type"))
        .def("field_class", [](LidarFrame& f, const std::string& name) { return f.field(name).field_class(); })
        .def_property_readonly("packet_count", &LidarFrame::packet_count)
        .def("get_first_valid_packet_timestamp", [](const LidarFrame& f) { return f.get_first_valid_packet_timestamp(); })
        .def("get_last_valid_packet_timestamp", [](const LidarFrame& f) { return f.get_last_valid_packet_timestamp(); })
        .def("get_min_valid_packet_timestamp", [](const LidarFrame& f) { return f.get_min_valid_packet_timestamp(); })
        .def("get_max_valid_packet_timestamp", [](const LidarFrame& f) { return f.get_max_valid_packet_timestamp(); })
        .def("del_field", [](LidarFrame& f, const std::string& n) { f.del_field(n); })
        .def_property_readonly("fields",
                               [](const LidarFrame& f) {
                                   std::vector<std::string> names;
                                   for (const auto& kv : f.fields()) names.push_back(kv.first);
                                   return names;
                               })
        .def_property_readonly("timestamp", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<uint64_t>(), {static_cast<py::ssize_t>(f.w)}, f.timestamp().data(), self);
        })
        .def_property_readonly("measurement_id", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<uint16_t>(), {static_cast<py::ssize_t>(f.w)}, f.measurement_id().data(), self);
        })
        .def_property_readonly("status", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<uint32_t>(), {static_cast<py::ssize_t>(f.w)}, f.status().data(), self);
        })
        .def_property_readonly("packet_timestamp", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<uint64_t>(), {static_cast<py::ssize_t>(f.packet_count())},
                             f.packet_timestamp().data(), self);
        })
        .def_property_readonly("alert_flags", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<uint8_t>(), {static_cast<py::ssize_t>(f.packet_count())},
                             f.alert_flags().data(), self);
        })
        // per-column 4x4 poses, (w, 4, 4) float64 view (python/src/cpp/client/lidar_frame.cpp:536-559)
        .def_property_readonly("body_to_world", [](py::object self) {
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<double>(), {static_cast<py::ssize_t>(f.w), py::ssize_t(4), py::ssize_t(4)},
                             f.body_to_world().get<double>(), self);
        })
        .def_property_readonly("pose", [](py::object self) {  // deprecated alias of body_to_world
            LidarFrame& f = self.cast<LidarFrame&>();
            return py::array(py::dtype::of<double>(), {static_cast<py::ssize_t>(f.w), py::ssize_t(4), py::ssize_t(4)},
                             f.body_to_world().get<double>(), self);
        })
        .def("get_first_valid_column", &LidarFrame::get_first_valid_column)
        .def("get_last_valid_column", &LidarFrame::get_last_valid_column)
        .def("__eq__", [](const LidarFrame& a, const LidarFrame& b) { return a == b; });

    py::class_<FrameBatcher>(m, "FrameBatcher")
        .def(py::init<const SensorInfo&>())
        .def("__call__", [](FrameBatcher& b, const LidarPacket& p, LidarFrame& f) { return b.batch(p, f); })
        .def("batch", [](FrameBatcher& b, const LidarPacket& p, LidarFrame& f) { return b.batch(p, f); })
        .def("flush", &FrameBatcher::flush)
        .def("reset", &FrameBatcher::reset)
        .def("set_max_cache_size", &FrameBatcher::set_max_cache_size)
        .def("get_max_cache_size", &FrameBatcher::get_max_cache_size)
        .def("batched_packets", &FrameBatcher::batched_packets)
        .def("dropped_packets", &FrameBatcher::dropped_packets);
    m.attr("ScanBatcher") = m.attr("FrameBatcher");
    m.attr("LidarScan") = m.attr("LidarFrame");

    m.def("get_field_types", [](const SensorInfo& info) { return get_field_types(info); }, py::arg("info"));
    m.def("get_field_types", [](UDPProfileLidar p) { return get_field_types(p); }, py::arg("udp_profile_lidar"));

    bind_lut<double>(m, "XYZLut");
    bind_lut<float>(m, "XYZLutFloat");

    // python/src/cpp/client/processing.cpp:527-608: any dtype, (h, w) or (h, w, n)
    m.def(
        "destagger",
        [](const SensorInfo& info, const py::array& field, bool inverse) {
            if (field.ndim() < 2) throw std::invalid_argument("Expected at least two dimensions for destagger");
            py::array src = py::array::ensure(field, py::array::c_style);
            const size_t h = static_cast<size_t>(src.shape(0)), w = static_cast<size_t>(src.shape(1));
            if (h != info.format.pixels_per_column || w != info.format.columns_per_frame ||
                h != info.format.pixel_shift_by_row.size() || src.size() == 0)
                throw std::invalid_argument("Image resolution must match SensorInfo.");
            size_t extra = 1;
            for (py::ssize_t i = 2; i < src.ndim(); ++i) extra *= static_cast<size_t>(src.shape(i));
            py::array out = pool_array(src.dtype(), std::vector<py::ssize_t>(src.shape(), src.shape() + src.ndim()));
            impl::destagger_bytes(src.data(), out.mutable_data(), h, w,
                                  static_cast<size_t>(src.itemsize()) * extra,
                                  info.format.pixel_shift_by_row, inverse, h, w);
            return out;
        },
        py::arg("info"), py::arg("field"), py::arg("inverse") = false);

    m.def(
        "frame_to_packets",
        [](const LidarFrame& f, std::shared_ptr<PacketFormat> pf, uint32_t init_id, uint64_t prod_sn) {
            return impl::frame_to_packets(f, std::move(pf), init_id, prod_sn);
        },
        py::arg("frame"), py::arg("packet_format"), py::arg("init_id") = 0, py::arg("prod_sn") = 0);

    // dewarp(points (H, W, 3), poses (W, 4, 4)) -> (H, W, 3): python/src/cpp/client/processing.cpp:132-164,
    // dtype dispatch :300-309 (float32 stays float32, everything else is computed in float64)
    m.def(
        "dewarp",
        [](const py::array& points, const py::array& poses) -> py::array {
            if (points.ndim() != 3 || points.shape(2) != 3 || poses.ndim() != 3 || poses.shape(1) != 4 ||
                poses.shape(2) != 4)
                throw std::invalid_argument("dewarp: expected points (H, W, 3) and poses (W, 4, 4)");
            if (points.shape(1) != poses.shape(0))
                throw std::runtime_error("Number of points per set must match number of poses");
            const size_t h = static_cast<size_t>(points.shape(0)), w = static_cast<size_t>(points.shape(1));
            auto po = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(poses);
            Poses pm(w, 16);
            if (w) std::memcpy(pm.data(), po.data(), w * 16 * sizeof(double));
            const std::vector<py::ssize_t> shape = {points.shape(0), points.shape(1), 3};
            if (points.dtype().is(py::dtype::of<float>())) {
                auto pt = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(points);
                py::array_t<float> out(shape);
                if (h * w) impl::dewarp_device(pt.data(), pm.data(), out.mutable_data(), false, h, w);
                return std::move(out);
            }
            auto pt = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(points);
            py::array_t<double> out(shape);
            if (h * w) impl::dewarp_device(pt.data(), pm.data(), out.mutable_data(), true, h, w);
            return std::move(out);
        },
        py::arg("points"), py::arg("poses"));

    // extension over the reference's Python surface: the C++ dewarp(LidarFrame, XYZLut, min_range,
    // max_range) with provenance (pose_util.h:456-485, impl/dewarp_impl.h:23-81)
    m.def(
        "dewarp_frame",
        [](const LidarFrame& frame, const XYZLutT<double>& lut, double min_range, double max_range) {
            std::vector<uint32_t> cols;
            std::vector<uint64_t> ts;
            auto pts = impl::dewarp_impl<double>(frame, lut, min_range, max_range, &cols, &ts);
            py::array_t<double> p({static_cast<py::ssize_t>(pts.size()), py::ssize_t(3)});
            if (!pts.empty()) std::memcpy(p.mutable_data(), pts.data(), pts.size() * 24);
            py::array_t<uint32_t> c(static_cast<py::ssize_t>(cols.size()));
            if (!cols.empty()) std::memcpy(c.mutable_data(), cols.data(), cols.size() * 4);
            py::array_t<uint64_t> t(static_cast<py::ssize_t>(ts.size()));
            if (!ts.empty()) std::memcpy(t.mutable_data(), ts.data(), ts.size() * 8);
            return py::make_tuple(p, c, t);
        },
        py::arg("frame"), py::arg("xyzlut"), py::arg("min_range"), py::arg("max_range"));

    // streaming pipeline for host-side packets (include/ouster/hip/frame_stream.h; extension over the
    // reference's Python surface).  The callback receives a dict of numpy VIEWS of pinned memory that
    // are valid only during the call: copy what must outlive it.
    {
        namespace oh = ouster::sdk::hip;
        py::class_<oh::FrameStream>(m, "FrameStream")
            // `info`: one SensorInfo, or a list of them (same data format; frame i of a batch belongs to sensor i % n)
            .def(py::init([](const py::object& info_or_list, py::function on_batch, uint32_t frames_per_batch,
                             uint32_t batches_in_flight, std::vector<std::string> planes,
                             std::vector<std::string> destaggered, bool xyz) {
                     std::vector<SensorInfo> infos;
                     if (py::isinstance<SensorInfo>(info_or_list)) infos.push_back(info_or_list.cast<const SensorInfo&>());
                     else infos = info_or_list.cast<std::vector<SensorInfo>>();
                     if (infos.empty()) throw std::invalid_argument("FrameStream: no sensor");
                     const SensorInfo& info = infos[0];
                     oh::StreamOptions opt;
                     opt.frames_per_batch = frames_per_batch;
                     opt.batches_in_flight = batches_in_flight;
                     opt.download_xyz = xyz;
                     opt.download_planes = planes;
                     opt.download_destaggered = destaggered;
                     opt.outputs.destagger = destaggered;
                     LidarFrame layout(info);
                     std::map<std::string, py::dtype> dts;
                     for (const auto& n : planes) dts.emplace(n, dtype_of(layout.field(n).tag()));
                     for (const auto& n : destaggered) dts.emplace(n, dtype_of(layout.field(n).tag()));
                     auto cb = [on_batch, dts](const oh::BatchResult& r) {
                         py::dict d;
                         const py::ssize_t n = r.n_frames, h = r.h, w = r.w;
                         auto view = [&](const py::dtype& dt, std::vector<py::ssize_t> shape, const void* p) {
                             return py::array(dt, std::move(shape), p, py::none());  // non-owning view
                         };
                         d["first_frame"] = r.first_frame;
                         d["n_frames"] = r.n_frames;
                         for (int k = 0; k < 2; ++k)
                             if (r.xyz[k])
                                 d[k == 0 ? "xyz" : "xyz2"] = view(py::dtype::of<float>(), {n, h, w, 3}, r.xyz[k]);
                         for (const auto& kv : r.planes) d[kv.first.c_str()] = view(dts.at(kv.first), {n, h, w}, kv.second);
                         for (const auto& kv : r.destaggered)
                             d[("destaggered:" + kv.first).c_str()] = view(dts.at(kv.first), {n, h, w}, kv.second);
                         if (r.timestamp) {
                             d["timestamp"] = view(py::dtype::of<uint64_t>(), {n, w}, r.timestamp);
                             d["measurement_id"] = view(py::dtype::of<uint16_t>(), {n, w}, r.measurement_id);
                             d["status"] = view(py::dtype::of<uint32_t>(), {n, w}, r.status);
                         }
                         on_batch(d);
                     };
                     return std::make_unique<oh::FrameStream>(infos, opt, cb);
                 }),
                 py::arg("info"), py::arg("on_batch"), py::arg("frames_per_batch") = 32,
                 py::arg("batches_in_flight") = 3, py::arg("planes") = std::vector<std::string>{},
                 py::arg("destaggered") = std::vector<std::string>{}, py::arg("xyz") = true)
            .def("push_frame",
                 [](oh::FrameStream& s, const std::vector<py::bytes>& packets) {
                     std::vector<std::string> keep(packets.begin(), packets.end());
                     std::vector<const uint8_t*> ptrs;
                     for (const auto& b : keep) ptrs.push_back(reinterpret_cast<const uint8_t*>(b.data()));
                     s.push_frame(ptrs);
                 })
            .def("push_packet", [](oh::FrameStream& s, const LidarPacket& p) { s.push_packet(p); })
            .def("push_packet", [](oh::FrameStream& s, size_t sensor, const LidarPacket& p) { s.push_packet(sensor, p); },
                 py::arg("sensor"), py::arg("packet"))
            .def("set_max_skew_frames", &oh::FrameStream::set_max_skew_frames)
            .def("finish", &oh::FrameStream::finish)
            .def_property_readonly("frames_pushed", &oh::FrameStream::frames_pushed)
            .def_property_readonly("frames_delivered", &oh::FrameStream::frames_delivered);
    }

    // OSF field planes (include/ouster/osf/osf.h): the container walk and the
    // entropy decoding on the host, the pixel work of every field of a batch in one GPU launch
    {
        namespace oo = ouster::sdk::osf;
        py::class_<oo::OsfFile>(m, "OsfFile")
            .def(py::init<const std::string&>())
            .def_property_readonly("version", &oo::OsfFile::version)
            .def_property_readonly("id", &oo::OsfFile::id)
            .def("metadata_types",
                 [](const oo::OsfFile& f) {
                     std::map<uint32_t, std::string> out;
                     for (const auto& e : f.metadata_entries()) out[e.id] = e.type;
                     return out;
                 })
            .def("sensor_metadata_json", &oo::OsfFile::sensor_metadata_json)
            .def("lidar_scan_streams", &oo::OsfFile::lidar_scan_streams)
            .def("messages", [](const oo::OsfFile& f) {
                py::list out;
                for (const auto& msg : f.messages())
                    out.append(py::make_tuple(msg.ts, msg.id,
                                              py::bytes(reinterpret_cast<const char*>(msg.buffer), msg.size)));
                return out;
            });
        // host half only (no GPU): what is staged for the device per field of a LidarScanMsg
        m.def("osf_stage_fields", [](const py::bytes& msg, size_t h, size_t w) {
            const std::string b = msg;
            oo::OsfFile::Message mm;
            mm.buffer = reinterpret_cast<const uint8_t*>(b.data());
            mm.size = b.size();
            const oo::LidarScanMsgView v = oo::LidarScanMsgView::parse(mm);
            py::list out;
            for (const auto& f : v.fields) {
                const oo::StagedField st = oo::stage_field(f, h, w);
                out.append(py::make_tuple(f.name, static_cast<int>(f.type), st.encoding, st.src_pixel_bytes,
                                          py::bytes(reinterpret_cast<const char*>(st.bytes.data()), st.bytes.size())));
            }
            return out;
        });
        py::class_<oo::OsfFrameDecoder>(m, "OsfFrameDecoder")
            .def(py::init<const SensorInfo&, int>(), py::arg("info"), py::arg("device") = -1)
            .def_property("device_unfilter", &oo::OsfFrameDecoder::device_unfilter, &oo::OsfFrameDecoder::set_device_unfilter)
            .def("decode", [](oo::OsfFrameDecoder& d, const std::vector<py::bytes>& msgs) {
                std::vector<std::string> keep(msgs.begin(), msgs.end());
                std::vector<oo::OsfFile::Message> mm(keep.size());
                for (size_t i = 0; i < keep.size(); ++i) {
                    mm[i].buffer = reinterpret_cast<const uint8_t*>(keep[i].data());
                    mm[i].size = keep[i].size();
                }
                return d.decode(mm);
            })
            .def("decode_fields", [](oo::OsfFrameDecoder& d, const std::vector<std::pair<py::bytes, int>>& blobs) {
                // [(encoded bytes, ChanFieldType value)] -> [bytes of the H x W plane]
                std::vector<std::string> keep;
                for (const auto& b : blobs) keep.emplace_back(b.first);
                std::vector<oo::EncodedField> ff(keep.size());
                for (size_t i = 0; i < keep.size(); ++i) {
                    ff[i].type = static_cast<ChanFieldType>(blobs[i].second);
                    ff[i].data = reinterpret_cast<const uint8_t*>(keep[i].data());
                    ff[i].size = keep[i].size();
                }
                std::vector<py::bytes> out;
                for (const auto& v : d.decode_fields(ff)) out.emplace_back(reinterpret_cast<const char*>(v.data()), v.size());
                return out;
            })
            .def("decode_device", [](oo::OsfFrameDecoder& d, const std::vector<py::bytes>& msgs) {
                std::vector<std::string> keep(msgs.begin(), msgs.end());
                std::vector<oo::OsfFile::Message> mm(keep.size());
                for (size_t i = 0; i < keep.size(); ++i) {
                    mm[i].buffer = reinterpret_cast<const uint8_t*>(keep[i].data());
                    mm[i].size = keep[i].size();
                }
                return d.decode_device(mm);
            });
        // planes that stay in HBM: pointers are plain integers, downloads go through numpy arrays
        py::class_<oo::OsfDeviceBatch>(m, "OsfDeviceBatch")
            .def_property_readonly("n_frames", &oo::OsfDeviceBatch::n_frames)
            .def_property_readonly("h", &oo::OsfDeviceBatch::h)
            .def_property_readonly("w", &oo::OsfDeviceBatch::w)
            .def_property_readonly("frame_ids", &oo::OsfDeviceBatch::frame_ids)
            .def("fields", [](const oo::OsfDeviceBatch& b) {
                py::dict out;
                for (const auto& f : b.fields()) out[py::str(f.first)] = static_cast<int>(f.second);
                return out;
            })
            .def("plane_ptr", [](const oo::OsfDeviceBatch& b, const std::string& n) {
                return reinterpret_cast<uintptr_t>(b.plane_device(n));
            })
            .def("destagger_ptr", [](oo::OsfDeviceBatch& b, const std::string& n) {
                return reinterpret_cast<uintptr_t>(b.destagger_device(n));
            })
            .def("cartesian_ptr", [](oo::OsfDeviceBatch& b, const XYZLut& lut, bool f64, const std::string& n) {
                return reinterpret_cast<uintptr_t>(b.cartesian_device(lut, f64, n));
            }, py::arg("lut"), py::arg("f64") = false, py::arg("range_field") = std::string(ChanField::RANGE))
            .def("download", [](const oo::OsfDeviceBatch& b, uintptr_t ptr, py::array out) {
                if (!(out.flags() & py::array::c_style) || !out.writeable())
                    throw std::invalid_argument("download needs a writeable C-contiguous array");
                b.download(reinterpret_cast<const void*>(ptr), out.mutable_data(), static_cast<size_t>(out.nbytes()));
            });
    }

    // every UDP datagram of a classic pcap as (payload bytes, destination port, capture time in ns): the Python face of
    // ouster::sdk::pcap::PcapReader (include/ouster/pcap/pcap.h), enough to feed a FrameBatcher from a capture
    m.def("read_pcap_udp", [](const std::string& path, uint64_t offset, int64_t max_packets) {
        ouster::sdk::pcap::PcapReader rd(path);
        if (offset) rd.seek(offset);
        py::list out;
        while (max_packets < 0 || static_cast<int64_t>(out.size()) < max_packets) {
            const size_t n = rd.next_packet();
            if (!n) break;
            const auto& info = rd.current_info();
            out.append(py::make_tuple(py::bytes(reinterpret_cast<const char*>(rd.current_data()), n), info.dst_port,
                                      static_cast<uint64_t>(info.timestamp.count()) * 1000ull));
        }
        return out;
    }, py::arg("path"), py::arg("offset") = 0, py::arg("max_packets") = -1);
    // a capture demultiplexed by sensor (include/ouster/pcap/indexed_pcap_reader.h): every UDP datagram as (sensor index or
    // None, payload bytes, destination port, capture time in ns, file offset) plus, per sensor, the file offsets at which
    // its frames start.  Sensors may share a port: datagrams are told apart by the ids in their packet headers.
    m.def("index_pcap", [](const std::string& path, const std::vector<SensorInfo>& infos, bool soft_id_check) {
        ouster::sdk::pcap::IndexedPcapReader rd(path, infos);
        py::list packets;
        while (size_t n = rd.next_packet()) {
            rd.update_index_for_current_packet();
            const auto& pi = rd.current_info();
            const auto idx = rd.sensor_idx_for_current_packet(soft_id_check);
            packets.append(py::make_tuple(idx ? py::object(py::int_(*idx)) : py::object(py::none()),
                                          py::bytes(reinterpret_cast<const char*>(rd.current_data()), n), pi.dst_port,
                                          static_cast<uint64_t>(pi.timestamp.count()) * 1000ull, pi.file_offset));
        }
        py::dict out;
        out["packets"] = packets;
        out["frame_offsets"] = rd.get_index().frame_indices;
        py::list ports;
        for (const SensorInfo& si : rd.sensor_info())
            ports.append(py::make_tuple(si.config.udp_port_lidar.value_or(0), si.config.udp_port_imu.value_or(0)));
        out["ports"] = ports;   // (lidar, imu) per sensor, guessed from the capture where the metadata does not name them
        return out;
    }, py::arg("path"), py::arg("sensor_info"), py::arg("soft_id_check") = false);
    m.def("default_lidar_to_sensor", [] { return mat_to(DEFAULT_LIDAR_TO_SENSOR); });
    m.def("default_beam_to_lidar_transform", [](const std::string& p) { return mat_to(default_beam_to_lidar_transform(p)); });
}
