/* Umbrella header (reference: src/include/starneig/starneig.h.in). */
#ifndef STARNEIG_AMD_STARNEIG_H
#define STARNEIG_AMD_STARNEIG_H
#include <starneig/error.h>
#include <starneig/node.h>
#include <starneig/expert.h>
#include <starneig/sep_sm.h>
#include <starneig/gep_sm.h>
#endif
