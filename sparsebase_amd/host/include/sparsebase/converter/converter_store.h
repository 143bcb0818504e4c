// Process-wide cache of converter singletons (reference: converter/converter_store.h:9-40).
#ifndef SPARSEBASE_CONVERTER_CONVERTER_STORE_H_
#define SPARSEBASE_CONVERTER_CONVERTER_STORE_H_
#include <memory>
#include <mutex>
#include <typeindex>
#include <unordered_map>

#include "sparsebase/converter/converter.h"

namespace sparsebase::converter {
class ConverterStore {
 public:
  static ConverterStore &GetStore() {
    static ConverterStore store;
    return store;
  }
  template <typename ConverterType>
  std::shared_ptr<ConverterType> get_converter() {
    std::lock_guard<std::mutex> lock(mu_);
    const std::type_index key(typeid(ConverterType));
    auto it = store_.find(key);
    if (it != store_.end())
      if (auto alive = it->second.lock()) return std::static_pointer_cast<ConverterType>(alive);
    auto fresh = std::make_shared<ConverterType>();
    store_[key] = fresh;
    return fresh;
  }
  ConverterStore(const ConverterStore &) = delete;
  ConverterStore &operator=(const ConverterStore &) = delete;

 private:
  ConverterStore() = default;
  std::mutex mu_;
  std::unordered_map<std::type_index, std::weak_ptr<Converter>> store_;
};
}  // namespace sparsebase::converter
#endif
