/* Caladan header stand-in (TEST INFRASTRUCTURE ONLY): everything lives in caladan_stub.h */
#pragma once
#include "../caladan_stub.h"
