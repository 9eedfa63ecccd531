#include "mpi_stub.h"
