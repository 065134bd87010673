#pragma once
#include "../../include/fhestring_hip.h"
#include "engine.h"
#include "strings.h"

struct fhs_ctx {
    fhs::Engine eng;
};
