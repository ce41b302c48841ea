// blokus.hip -- placeholder until the Blokus kernels land (see DESIGN.md build order)
#include "crl_common.hpp"
void crl_blokus_free(void *tables) { (void)tables; }
