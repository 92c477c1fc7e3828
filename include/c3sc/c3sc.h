/* c3sc.h -- umbrella header (mirrors src/c3sc.h:40-51) */
#ifndef C3SC_H
#define C3SC_H
#include "bellman.h"
#include "boundary.h"
#include "dynamics.h"
#include "hashgrid.h"
#include "nodeutil.h"
#include "util.h"
#include "valuefunc.h"
#endif
