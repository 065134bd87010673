// Lazy DAG engine over the batched-PBS context (filled in by engine.cpp).
#pragma once
#include "context.h"

namespace fhs {

class Engine {
  public:
    Context ctx;
    int on_key_loaded() { return 0; }
    void shutdown() { ctx.shutdown(); }
};

}  // namespace fhs
