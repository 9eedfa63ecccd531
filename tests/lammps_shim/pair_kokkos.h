#include "kokkos_shim.h"
