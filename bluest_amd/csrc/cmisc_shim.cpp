// cmisc_shim.cpp -- pybind11 module `_cmisc_bluest` with exactly the five function names, argument orders and
// accumulate-in-place semantics of the reference's native module (bluest/cmisc.cpp:99-110), forwarding to the C-ABI
// of libbluest_hip.so.  Dropping this .so on sys.path ahead of the reference's own makes bluest/misc.py:11
//     from _cmisc_bluest import assemble_psi_c,objectiveK_c,gradK_c,hessKQ_c,cleanupK_c
// run on the GPU with no source change (level 0).  No compute happens here.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <stdexcept>
#include <string>

#include "bluest_hip.h"

namespace py = pybind11;
using darr = py::array_t<double, py::array::c_style>;
using iarr = py::array_t<long int, py::array::c_style>;

static void chk(int rc)
{
    if (rc != BLUEST_OK) throw std::runtime_error(std::string("libbluest_hip: ") + bluest_last_error());
}

static void assemble_psi_c(darr psi, const int N, const int k, const int Lk, const iarr groupsk, const darr invcovsk)
{ chk(bluest_assemble_psi((double *)psi.data(0), N, k, Lk, (const int64_t *)groupsk.data(0), invcovsk.data(0))); }

static void objectiveK_f64(darr PHI, const int N, const int k, const int Lk, const darr mk, const iarr groupsk, const darr invcovsk)
{ chk(bluest_objectiveK_f64((double *)PHI.data(0), N, k, Lk, mk.data(0), (const int64_t *)groupsk.data(0), invcovsk.data(0))); }

static void objectiveK_i64(darr PHI, const int N, const int k, const int Lk, const iarr mk, const iarr groupsk, const darr invcovsk)
{ chk(bluest_objectiveK_i64((double *)PHI.data(0), N, k, Lk, (const int64_t *)mk.data(0), (const int64_t *)groupsk.data(0), invcovsk.data(0))); }

static void gradK_c(darr grad, const int k, const int Lk, const iarr groupsk, const darr invcovsk, const darr invPHI_0)
{ chk(bluest_gradK((double *)grad.data(0), k, Lk, (const int64_t *)groupsk.data(0), invcovsk.data(0), invPHI_0.data(0), (int)invPHI_0.size())); }

static void cleanupK_c(darr X, const int k, const int Lk, const iarr groupsk, const darr invcovsk, const darr invPHI_0)
{ chk(bluest_cleanupK((double *)X.data(0), k, Lk, (const int64_t *)groupsk.data(0), invcovsk.data(0), invPHI_0.data(0), (int)invPHI_0.size())); }

static void hessKQ_c(darr hess, const int N, const int k, const int q, const int Lk, const int Lq, const iarr groupsk, const iarr groupsq,
                     const darr invcovsk, const darr invcovsq, const darr invPHI)
{
    chk(bluest_hessKQ((double *)hess.data(0), N, k, q, Lk, Lq, (const int64_t *)groupsk.data(0), (const int64_t *)groupsq.data(0),
                      invcovsk.data(0), invcovsq.data(0), invPHI.data(0)));
}

PYBIND11_MODULE(_cmisc_bluest, m)
{
    m.doc() = "GPU (MI355X) replacement for bluest's _cmisc_bluest, forwarding to libbluest_hip.so";
    m.def("assemble_psi_c", &assemble_psi_c);
    m.def("objectiveK_c", &objectiveK_f64);
    m.def("objectiveK_c", &objectiveK_i64);
    m.def("gradK_c", &gradK_c);
    m.def("hessKQ_c", &hessKQ_c);
    m.def("cleanupK_c", &cleanupK_c);
}
