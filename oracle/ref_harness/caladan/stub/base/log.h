/* Caladan base/log.h stand-in (TEST INFRASTRUCTURE ONLY; included by the clients inside extern "C": plain C) */
#pragma once
#include <stdio.h>
#include <stdlib.h>
#define log_emerg(fmt, ...) do { if (getenv("REF_CLIENT_VERBOSE")) fprintf(stderr, fmt "\n", ##__VA_ARGS__); } while (0)
#define log_info log_emerg
#define log_err log_emerg
#define panic(fmt, ...) do { fprintf(stderr, "panic: " fmt "\n", ##__VA_ARGS__); abort(); } while (0)
