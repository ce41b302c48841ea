/* blokus_oracle.c -- placeholder, filled in below */
#include "crl_oracle.h"
