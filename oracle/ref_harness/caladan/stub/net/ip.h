/* Caladan net/ip.h stand-in (TEST INFRASTRUCTURE ONLY; included by the clients inside extern "C": plain C) */
#pragma once
#include <stdint.h>
#define MAKE_IP_ADDR(a, b, c, d) ((((uint32_t)(a)) << 24) | (((uint32_t)(b)) << 16) | (((uint32_t)(c)) << 8) | ((uint32_t)(d)))
