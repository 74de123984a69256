// Test infrastructure: a pybind11 module that implements ONLY the pickle contract of the reference's
// `particlesfm` module (point_trajectory/optimize/src/bindings.cc:33-75 over trajectory_base.cpp:39-53,95-107):
//   TrajectorySet.__setstate__(std::map<int, py::dict>)  ->  Trajectory(py::dict) with the three casts
//   dict["frame_ids"] -> std::vector<int>, dict["locations"] -> std::vector<2-vector of double>,
//   dict["labels"] -> std::vector<bool>,   and as_dict() back.
// It is written here from that contract (Eigen is not in this image: V2D = Eigen::Vector2d is stood in for by
// std::array<double, 2>, whose pybind caster likewise takes any length-2 sequence / (2,) ndarray row).  The test
// builds it with g++ + the pip pybind11 headers, installs it at point_trajectory/optimize/build/ in a scratch tree
// WITHOUT this package on the path, and loads the DEFAULT track.npy this package writes.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <map>
#include <vector>

namespace py = pybind11;
using V2 = std::array<double, 2>;

struct Trajectory {
    std::vector<int> times;
    std::vector<V2> xys;
    std::vector<bool> labels;
    Trajectory() {}
    explicit Trajectory(py::dict d)
    {
        if (d.contains("frame_ids")) times = d["frame_ids"].cast<std::vector<int>>();
        if (d.contains("locations")) xys = d["locations"].cast<std::vector<V2>>();
        if (d.contains("labels")) labels = d["labels"].cast<std::vector<bool>>();
    }
    py::dict as_dict() const
    {
        py::dict o;
        o["frame_ids"] = times;
        o["locations"] = xys;
        o["labels"] = labels;
        return o;
    }
};

struct TrajectorySet {
    std::map<int, Trajectory> trajs;
    TrajectorySet() {}
    explicit TrajectorySet(std::map<int, py::dict> in)
    {
        for (auto& kv : in) trajs.insert(std::make_pair(kv.first, Trajectory(kv.second)));
    }
    std::map<int, py::dict> as_dict() const
    {
        std::map<int, py::dict> o;
        for (auto& kv : trajs) o[kv.first] = kv.second.as_dict();
        return o;
    }
};

PYBIND11_MODULE(particlesfm, m)
{
    py::class_<Trajectory>(m, "Trajectory")
        .def(py::init<py::dict>())
        .def("as_dict", &Trajectory::as_dict)
        .def(py::pickle([](const Trajectory& t) { return t.as_dict(); }, [](const py::dict& d) { return Trajectory(d); }));
    py::class_<TrajectorySet>(m, "TrajectorySet")
        .def(py::init<>())
        .def(py::init<std::map<int, py::dict>>())
        .def("as_dict", &TrajectorySet::as_dict)
        .def(py::pickle([](const TrajectorySet& s) { return s.as_dict(); },
                        [](const std::map<int, py::dict>& d) { return TrajectorySet(d); }));
}
