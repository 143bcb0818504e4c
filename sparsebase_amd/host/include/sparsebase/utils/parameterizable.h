// Hyper-parameter carriers (reference: utils/parameterizable.h:11-21): operators keep their settings in a
// struct derived from Parameters and hand a pointer to it to the implementation function they dispatch to.
#ifndef SPARSEBASE_UTILS_PARAMETERIZABLE_H_
#define SPARSEBASE_UTILS_PARAMETERIZABLE_H_
#include <memory>

namespace sparsebase::utils {

struct Parameters {  // polymorphic on purpose: implementation functions down-cast it
  virtual ~Parameters() = default;
};

class Parameterizable {  // base of every operator that owns one Parameters object
 protected:
  std::unique_ptr<Parameters> params_;

 public:
  virtual ~Parameterizable() = default;
  // read-only view for callers that only want to inspect the settings
  const Parameters *parameters() const { return params_.get(); }
};

}  // namespace sparsebase::utils
#endif
