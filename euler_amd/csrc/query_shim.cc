// euler::Query / euler::QueryProxy for the hot-path TF kernels (see
// include/euler_query.h): a recogniser for the query shapes those kernels
// generate, routed through the plugin ops of op_framework.cc.  No GQL parser.
#include "euler_query.h"

#include <string.h>

#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <thread>

namespace euler {
inline namespace gpu_abi {

namespace {

void LogError(const std::string& msg) {
  fprintf(stderr, "[euler_gpu] ERROR %s\n", msg.c_str());
}

// ---- tokens of the recognised shapes: identifiers, integers, ( ) , .
struct Step {
  std::string fn;                  // v, sampleNB, outV, sampleN, as, has, ...
  std::vector<std::string> args;
};

bool Tokenise(const std::string& q, std::vector<Step>* steps) {
  size_t i = 0;
  const size_t n = q.size();
  auto skip = [&] { while (i < n && (q[i] == ' ' || q[i] == '\t' || q[i] == '\n')) ++i; };
  while (true) {
    skip();
    if (i >= n) break;
    Step st;
    while (i < n && (isalnum((unsigned char)q[i]) || q[i] == '_')) st.fn.push_back(q[i++]);
    skip();
    if (st.fn.empty() || i >= n || q[i] != '(') return false;
    ++i;
    std::string cur;
    int depth = 1;
    while (i < n && depth > 0) {
      const char c = q[i++];
      if (c == '(') { ++depth; cur.push_back(c); }
      else if (c == ')') { if (--depth > 0) cur.push_back(c); }
      else if (c == ',' && depth == 1) { st.args.push_back(cur); cur.clear(); }
      else if (c != ' ' && c != '\t') cur.push_back(c);
    }
    if (depth != 0) return false;
    if (!cur.empty() || !st.args.empty()) st.args.push_back(cur);
    steps->push_back(st);
    skip();
    if (i < n) {
      if (q[i] != '.') return false;
      ++i;
    }
  }
  return !steps->empty();
}

bool RunOp(const NodeDef& nd, OpKernelContext* ctx) {
  OpKernel* kernel = nullptr;
  if (CreateOpKernel(nd.op, &kernel) != 0) {
    LogError("no kernel registered for " + nd.op);
    return false;
  }
  kernel->Compute(nd, ctx);
  Tensor* t = nullptr;
  return ctx->tensor(OutputName(nd, 0), &t) == 0;     // ops leave no output on error
}

// ---- the proxy's query threads (client/query_proxy.cc:205-210: a pool of 8)
class Pool {
 public:
  static Pool* Get() {
    static Pool* p = new Pool(8);      // leaked: threads may outlive static destructors
    return p;
  }
  void Submit(std::function<void()> fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back(std::move(fn));
    }
    cv_.notify_one();
  }

 private:
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i)
      std::thread([this] {
        for (;;) {
          std::function<void()> fn;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return !q_.empty(); });
            fn = std::move(q_.front());
            q_.pop_front();
          }
          fn();
        }
      }).detach();
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
};

std::mutex g_proxy_mu;
QueryProxy* g_proxy = nullptr;
euler_gpu_graph* g_proxy_graph = nullptr;

}  // namespace

Query::Query(const std::string& gremlin)
    : ctx_(std::make_shared<OpKernelContext>()), gremlin_(gremlin) {}
Query::~Query() {}

Tensor* Query::AllocInput(const std::string& name, const TensorShape& shape,
                          const DataType& type) {
  Tensor* t = nullptr;
  if (ctx_->Allocate(name, shape, type, &t) != 0) {
    LogError("AllocInput: tensor '" + name + "' exists");
    return nullptr;
  }
  return t;
}

std::unordered_map<std::string, Tensor*> Query::GetResult(
    const std::vector<std::string>& result_names) {
  std::unordered_map<std::string, Tensor*> out;
  for (const std::string& name : result_names) {
    Tensor* t = nullptr;
    out[name] = ctx_->tensor(name, &t) == 0 ? t : nullptr;
  }
  return out;
}

Tensor* Query::GetResult(const std::string& result_name) {
  Tensor* t = nullptr;
  return ctx_->tensor(result_name, &t) == 0 ? t : nullptr;
}

QueryProxy* QueryProxy::GetInstance() {
  std::lock_guard<std::mutex> lk(g_proxy_mu);
  if (g_proxy == nullptr && (g_proxy_graph != nullptr || euler_gpu_default_graph() != nullptr))
    g_proxy = new QueryProxy();
  if (g_proxy == nullptr) LogError("Init failed");      // query_proxy.h:61-66
  return g_proxy;
}

bool QueryProxy::Init(euler_gpu_graph* graph) {
  std::lock_guard<std::mutex> lk(g_proxy_mu);
  g_proxy_graph = graph;
  if (g_proxy == nullptr && graph != nullptr) g_proxy = new QueryProxy();
  return graph != nullptr;
}

void QueryProxy::SetSeed(uint64_t seed) { OpKernelContext::SetProcessSeed(seed); }

bool QueryProxy::Execute(Query* query) {
  OpKernelContext* ctx = query->ctx_.get();
  {
    std::lock_guard<std::mutex> lk(g_proxy_mu);
    if (g_proxy_graph != nullptr) ctx->SetGraph(g_proxy_graph);
  }
  std::vector<Step> steps;
  if (!Tokenise(query->gremlin_, &steps)) {
    LogError("unsupported query (not one of the hot-path shapes): " + query->gremlin_);
    return false;
  }
  for (const Step& st : steps)
    if (st.fn == "has" || st.fn == "has_key" || st.fn == "has_label") {
      LogError("conditions need the attribute index, which this backend does not build: " +
               query->gremlin_);
      return false;
    }
  size_t i = 0;
  // sampleN(node_type, count).as(alias)           tf_euler/kernels/sample_node_op.cc:63-72
  if (steps[0].fn == "sampleN") {
    if (steps.size() != 2 || steps[0].args.size() != 2 || steps[1].fn != "as" ||
        steps[1].args.size() != 1) {
      LogError("unsupported sampleN query: " + query->gremlin_);
      return false;
    }
    NodeDef nd{steps[1].args[0], "API_SAMPLE_NODE", {steps[0].args[0], steps[0].args[1]}, {}};
    return RunOp(nd, ctx);
  }
  if (steps[0].fn != "v" || steps[0].args.size() != 1) {
    LogError("unsupported query: " + query->gremlin_);
    return false;
  }
  std::string ids = steps[0].args[0];            // name of the current id tensor
  for (i = 1; i < steps.size();) {
    const Step& st = steps[i];
    std::vector<std::string> post;
    size_t j = i + 1;
    // order_by(field, asc|desc) / limit(k) between the step and its alias:
    // DAGNodeProto.post_process entries (parser: "order_by id desc", "limit 3")
    for (; j < steps.size() && (steps[j].fn == "order_by" || steps[j].fn == "limit"); ++j) {
      std::string p = steps[j].fn;
      for (size_t a = 0; a < steps[j].args.size(); ++a)
        if (!(steps[j].fn == "order_by" && a == 1 && steps[j].args[a] == "asc"))
          p += " " + steps[j].args[a];
      post.push_back(p);
    }
    if (j >= steps.size() || steps[j].fn != "as" || steps[j].args.size() != 1) {
      LogError("every traversal step must be aliased with .as(name): " + query->gremlin_);
      return false;
    }
    const std::string alias = steps[j].args[0];
    if (st.fn == "sampleNB" && st.args.size() == 3) {
      // v(...).sampleNB(edge_types, count, default_node): API_SAMPLE_NB; the literal
      // default node travels as the input string and is ignored (sample_neighbor_op.cc:134)
      NodeDef nd{alias, "API_SAMPLE_NB", {ids, st.args[0], st.args[1], st.args[2]}, post};
      if (!RunOp(nd, ctx)) return false;
    } else if (st.fn == "outV" && st.args.size() == 1) {
      NodeDef nd{alias, "API_GET_NB_NODE", {ids, st.args[0]}, post};
      if (!RunOp(nd, ctx)) return false;
    } else {
      LogError("unsupported traversal step '" + st.fn + "': " + query->gremlin_);
      return false;
    }
    ids = alias + ":1";                            // the next step starts from these ids
    i = j + 1;
  }
  return true;
}

std::unordered_map<std::string, Tensor*> QueryProxy::RunGremlin(
    Query* query, const std::vector<std::string>& result_names) {
  Execute(query);
  return query->GetResult(result_names);
}

void QueryProxy::RunAsyncGremlin(Query* query, DoneCallback callback) {
  Pool::Get()->Submit([this, query, callback] {
    Execute(query);
    callback();          // errors were logged; the results are simply missing (Q12)
  });
}

}  // namespace gpu_abi
}  // namespace euler

extern "C" {

void euler_query_set_seed(uint64_t seed) { euler::QueryProxy::SetSeed(seed); }
void euler_query_set_graph(euler_gpu_graph* graph) { (void)euler::QueryProxy::Init(graph); }

int64_t euler_query_run(const char* gremlin, int32_t n_inputs, const char* const* names,
                        const int32_t* dtypes, const int64_t* counts,
                        const void* const* data, const char* result_name, void* out,
                        int64_t capacity) {
  using namespace euler;
  QueryProxy* proxy = QueryProxy::GetInstance();
  if (proxy == nullptr) return -1;
  Query* query = new Query(gremlin);
  for (int32_t i = 0; i < n_inputs; ++i) {
    Tensor* t = counts[i] < 0
        ? query->AllocInput(names[i], {}, (DataType)dtypes[i])          // scalar
        : query->AllocInput(names[i], {(size_t)counts[i]}, (DataType)dtypes[i]);
    if (t == nullptr) { delete query; return -1; }
    memcpy(t->Raw<char>(), data[i], t->TotalBytes());
  }
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  proxy->RunAsyncGremlin(query, [&] {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv.notify_one();
  });
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
  Tensor* res = query->GetResult(result_name);
  int64_t rc = -1;
  if (res != nullptr) {
    rc = (int64_t)res->TotalBytes();
    if (rc > capacity) rc = -2;
    else memcpy(out, res->Raw<char>(), (size_t)rc);
  }
  delete query;
  return rc;
}

}  // extern "C"
