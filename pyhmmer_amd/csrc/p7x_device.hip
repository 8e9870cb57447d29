#include "p7x_internal.hpp"
extern "C" void p7x_oprofile_destroy(p7x_oprofile *om) { delete om; }
