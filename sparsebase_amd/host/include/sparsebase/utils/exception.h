// Exception types of the host layer; same names and meaning as the reference's
// utils/exception.h:23-190 so call sites and tests can catch the same classes.
#ifndef SPARSEBASE_UTILS_EXCEPTION_H_
#define SPARSEBASE_UTILS_EXCEPTION_H_
#include <exception>
#include <string>
#include <typeindex>
#include <vector>

namespace sparsebase::utils {

class Exception : public std::exception {
 public:
  Exception() = default;
  explicit Exception(std::string msg) : msg_(std::move(msg)) {}
  const char *what() const noexcept override { return msg_.c_str(); }

 protected:
  std::string msg_;
};

class InvalidDataMember : public Exception {
 public:
  InvalidDataMember(const std::string &f, const std::string &dm)
      : Exception("Format " + f + " does not have " + dm + " as a data member.") {}
};

class ReaderException : public Exception {
 public:
  explicit ReaderException(const std::string &msg) : Exception(msg) {}
};

class WriterException : public Exception {
 public:
  explicit WriterException(const std::string &msg) : Exception(msg) {}
};

class TypeException : public Exception {
 public:
  explicit TypeException(const std::string &msg) : Exception(msg) {}
  TypeException(const std::string &type1, const std::string &type2)
      : Exception("Object is of type " + type1 + " not " + type2) {}
};

class ConversionException : public Exception {
 public:
  ConversionException(const std::string &type1, const std::string &type2)
      : Exception("Cannot convert from " + type1 + " to " + type2) {}
};

class FeatureException : public Exception {
 public:
  FeatureException(const std::string &feature, const std::string &extractor)
      : Exception("ERROR! " + feature + " is not registered in " + extractor + "!") {}
};

class FeatureParamsException : public FeatureException {
 public:
  using FeatureException::FeatureException;
};

// Thrown when an operator has no implementation for the input format and the
// caller passed convert_input == false (function_matcher_mixin.h:196-202).
template <typename KeyType>
class DirectExecutionNotAvailableException : public Exception {
 public:
  DirectExecutionNotAvailableException(const KeyType &used, const std::vector<KeyType> &available)
      : used_(used), available_(available) {
    msg_ = "Could not find a function for the given format that does not require conversion";
  }
  KeyType used_format() const { return used_; }
  std::vector<KeyType> available_formats() const { return available_; }

 protected:
  KeyType used_;
  std::vector<KeyType> available_;
};

class FunctionNotFoundException : public Exception {
 public:
  explicit FunctionNotFoundException(const std::string &msg) : Exception(msg) {}
};

class NoConverterException : public Exception {
 public:
  NoConverterException() : Exception("Attempting to use a format that does not have a converter") {}
};

template <typename T>
class AttemptToReset : public Exception {
 public:
  AttemptToReset() : Exception("Attempt to reset a once-settable value") {}
};

// Device errors: the HIP analogue of the reference's CUDADeviceException (:178).
class HIPDeviceException : public Exception {
 public:
  HIPDeviceException(int available_devices, int requested_device)
      : Exception("Attempting to use HIP device " + std::to_string(requested_device) + " while only " +
                  std::to_string(available_devices) + " HIP devices are available") {}
  explicit HIPDeviceException(const std::string &msg) : Exception(msg) {}
};

}  // namespace sparsebase::utils
#endif
