#include "lammps_shim.h"
