#define LAMMPS_VERSION "shim"
