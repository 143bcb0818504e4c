// Parameters / Parameterizable (reference: utils/parameterizable.h:11-21).
#ifndef SPARSEBASE_UTILS_PARAMETERIZABLE_H_
#define SPARSEBASE_UTILS_PARAMETERIZABLE_H_
#include <memory>

namespace sparsebase::utils {
struct Parameters {
  virtual ~Parameters() = default;
};
class Parameterizable {
 public:
  virtual ~Parameterizable() = default;

 protected:
  std::unique_ptr<Parameters> params_;
};
}  // namespace sparsebase::utils
#endif
