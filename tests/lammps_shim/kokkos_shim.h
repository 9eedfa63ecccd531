// TEST INFRASTRUCTURE ONLY: the few declarations of Kokkos and of the LAMMPS KOKKOS package that
// pair_allegro_hip_kokkos.cpp touches (views, dual views, the execution space's stream, AtomKokkos sync/modified masks,
// NeighListKokkos' device table), just enough to compile the class and drive it with the LAMMPS call sequence.
// "Device" memory is plain host memory by default (the host-emulation library treats device pointers as host pointers);
// with -DSHIM_HIP it is hipMalloc'ed memory, so the same driver runs the class against the real liballegro_hip.so on a GPU.
// Not Kokkos, not LAMMPS, not shipped.
#pragma once
#include "lammps_shim.h"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <type_traits>

#ifdef SHIM_HIP
#include <hip/hip_runtime_api.h>
#define SHIM_CHECK(e) do { if ((e) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); std::abort(); } } while (0)
#endif

namespace Kokkos {
struct HostSpace {};
struct LayoutRight {};
struct LayoutLeft {};
struct ShimDevice {      // stands for Kokkos::HIP
  void *hip_stream() const { return nullptr; }
  void fence() const {
#ifdef SHIM_HIP
    SHIM_CHECK(hipDeviceSynchronize());
#endif
  }
};
namespace shim {
template <class Space> struct Mem {
  static void *alloc(size_t n) { return std::calloc(n ? n : 1, 1); }
  static void free(void *p) { std::free(p); }
};
#ifdef SHIM_HIP
template <> struct Mem<ShimDevice> {
  static void *alloc(size_t n) { void *p = nullptr; SHIM_CHECK(hipMalloc(&p, n ? n : 1)); SHIM_CHECK(hipMemset(p, 0, n ? n : 1)); return p; }
  static void free(void *p) { (void) hipFree(p); }
};
#endif
inline void copy(void *dst, const void *src, size_t n) {
#ifdef SHIM_HIP
  SHIM_CHECK(hipMemcpy(dst, src, n, hipMemcpyDefault));
#else
  std::memcpy(dst, src, n);
#endif
}
template <class D> struct Dims;                                        // rank, element type, compile-time inner extent
template <class T> struct Dims<T *> { using type = T; static constexpr int rank = 1; static constexpr size_t inner = 1; };
template <class T> struct Dims<T **> { using type = T; static constexpr int rank = 2; static constexpr size_t inner = 0; };
template <class T, size_t N> struct Dims<T *[N]> { using type = T; static constexpr int rank = 2; static constexpr size_t inner = N; };
template <class... P> struct Pick { using space = ShimDevice; static constexpr bool left = false; };
template <class... P> struct Pick<HostSpace, P...> { using space = HostSpace; static constexpr bool left = Pick<P...>::left; };
template <class... P> struct Pick<ShimDevice, P...> { using space = ShimDevice; static constexpr bool left = Pick<P...>::left; };
template <class... P> struct Pick<LayoutLeft, P...> { using space = typename Pick<P...>::space; static constexpr bool left = true; };
template <class... P> struct Pick<LayoutRight, P...> { using space = typename Pick<P...>::space; static constexpr bool left = false; };
}    // namespace shim

template <class DataType, class... P> class View {
 public:
  using D = shim::Dims<DataType>;
  using T = typename D::type;
  using Space = typename shim::Pick<P...>::space;
  using HostMirror = View<DataType, HostSpace, typename std::conditional<shim::Pick<P...>::left, LayoutLeft, LayoutRight>::type>;
  View() = default;
  View(const char *, size_t n0, size_t n1 = 0) { n_[0] = n0; n_[1] = D::rank == 1 ? 1 : (D::inner ? D::inner : n1); mem_.reset((T *) shim::Mem<Space>::alloc(n_[0] * n_[1] * sizeof(T)), shim::Mem<Space>::free); }
  T *data() const { return mem_.get(); }
  size_t extent(int r) const { return r < D::rank ? n_[r] : 1; }
  size_t stride(int r) const { return shim::Pick<P...>::left ? (r == 0 ? 1 : n_[0]) : (r == 0 ? n_[1] : 1); }
  size_t size() const { return n_[0] * n_[1]; }
  // element access: meaningful only for host-accessible memory (the drivers fill host mirrors, the class reads h_engvir)
  T &operator()(size_t i) const { return mem_.get()[i]; }
  T &operator()(size_t i, size_t j) const { return mem_.get()[i * stride(0) + j * stride(1)]; }
 private:
  std::shared_ptr<T> mem_;
  size_t n_[2] = {0, 1};
};
template <class V> typename V::HostMirror create_mirror_view(const V &v) { return typename V::HostMirror("mirror", v.extent(0), v.extent(1)); }
template <class A, class B> void deep_copy(const A &dst, const B &src) { shim::copy(dst.data(), src.data(), src.size() * sizeof(typename B::T)); }
template <class E, class A, class B> void deep_copy(const E &, const A &dst, const B &src) { deep_copy(dst, src); }

template <class DataType, class... P> class DualView {
 public:
  using t_dev = View<DataType, P...>;
  using t_host = typename t_dev::HostMirror;
  t_dev d_view; t_host h_view;
  bool host_dirty = false, dev_dirty = false;
  DualView() = default;
  DualView(const char *n, size_t n0, size_t n1 = 0) : d_view(n, n0, n1), h_view(n, n0, n1) {}
  template <class S> typename std::conditional<std::is_same<S, HostSpace>::value, t_host, t_dev>::type view() const {
    if constexpr (std::is_same<S, HostSpace>::value) return h_view; else return d_view;
  }
  template <class S> void modify() { if (std::is_same<S, HostSpace>::value) host_dirty = true; else dev_dirty = true; }
  template <class S> void sync() {
    if (std::is_same<S, HostSpace>::value) { if (dev_dirty) { deep_copy(h_view, d_view); dev_dirty = false; } }
    else if (host_dirty) { deep_copy(d_view, h_view); host_dirty = false; }
  }
};
}    // namespace Kokkos

namespace LAMMPS_NS {
typedef Kokkos::ShimDevice LMPDeviceType;
typedef Kokkos::HostSpace LMPHostType;
typedef double X_FLOAT;
typedef double F_FLOAT;
typedef double E_FLOAT;
template <class D> struct ExecutionSpaceFromDevice { static const ExecutionSpace space = Device; };
template <> struct ExecutionSpaceFromDevice<LMPHostType> { static const ExecutionSpace space = Host; };
enum { FULL = 1u, HALFTHREAD = 2u, HALF = 4u };
enum { X_MASK = 1, V_MASK = 2, F_MASK = 4, TAG_MASK = 8, TYPE_MASK = 16, ENERGY_MASK = 0x10000, VIRIAL_MASK = 0x20000 };

struct DAT {
  typedef Kokkos::DualView<X_FLOAT *[3], Kokkos::LayoutRight, LMPDeviceType> tdual_x_array;
  typedef Kokkos::DualView<F_FLOAT *[3], Kokkos::LayoutRight, LMPDeviceType> tdual_f_array;
  typedef Kokkos::DualView<int *, LMPDeviceType> tdual_int_1d;
  typedef Kokkos::DualView<E_FLOAT *, LMPDeviceType> tdual_efloat_1d;
  typedef Kokkos::DualView<int **, Kokkos::LayoutLeft, LMPDeviceType> tdual_neighbors_2d;     // column-major on a GPU build
};
template <class DeviceType> struct ArrayTypes {
  typedef DAT::tdual_x_array::t_dev t_x_array_randomread;
  typedef DAT::tdual_f_array::t_dev t_f_array;
  typedef DAT::tdual_int_1d::t_dev t_int_1d_randomread;
  typedef DAT::tdual_efloat_1d::t_dev t_efloat_1d;
  typedef DAT::tdual_neighbors_2d::t_dev t_neighbors_2d;
};

class AtomKokkos : public Atom {
 public:
  DAT::tdual_x_array k_x; DAT::tdual_f_array k_f; DAT::tdual_int_1d k_type, k_tag;
  unsigned synced_to_device = 0, modified_on_device = 0;     // what the pair style asked for (the driver asserts on these)
  void sync(ExecutionSpace space, unsigned mask) {
    if (space == Device) {
      synced_to_device |= mask;
      if (mask & X_MASK) k_x.sync<LMPDeviceType>();
      if (mask & F_MASK) k_f.sync<LMPDeviceType>();
      if (mask & TYPE_MASK) k_type.sync<LMPDeviceType>();
      if (mask & TAG_MASK) k_tag.sync<LMPDeviceType>();
    } else {
      if (mask & X_MASK) k_x.sync<LMPHostType>();
      if (mask & F_MASK) k_f.sync<LMPHostType>();
    }
  }
  void modified(ExecutionSpace space, unsigned mask) {
    if (space == Device) { modified_on_device |= mask; if (mask & F_MASK) k_f.modify<LMPDeviceType>(); }
    else { if (mask & X_MASK) k_x.modify<LMPHostType>(); if (mask & F_MASK) k_f.modify<LMPHostType>(); if (mask & TYPE_MASK) k_type.modify<LMPHostType>(); if (mask & TAG_MASK) k_tag.modify<LMPHostType>(); }
  }
};
class MemoryKokkos : public Memory {
 public:
  template <class DV, class T> void create_kokkos(DV &k, T *&host, int n, const char *name) { k = DV(name, (size_t) n); host = k.h_view.data(); }
  template <class DV, class T> void destroy_kokkos(DV &k, T *&host) { k = DV(); host = nullptr; }
};
template <class DeviceType> class NeighListKokkos : public NeighList {
 public:
  typename ArrayTypes<DeviceType>::t_neighbors_2d d_neighbors;
  typename ArrayTypes<DeviceType>::t_int_1d_randomread d_ilist, d_numneigh;
};
class KokkosLMP { public: int neighflag = HALF; };
}    // namespace LAMMPS_NS
