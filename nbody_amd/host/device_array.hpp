// device_array.hpp - a move-only array of doubles in HBM (hipMalloc / hipFree), for state that is
// generated on the device and handed from one part of the host code to another without passing
// through host memory.
#pragma once
#include <cstddef>
#include <vector>

namespace nbh {

class DeviceArray {
   public:
    DeviceArray() = default;
    explicit DeviceArray(std::size_t count);          // uninitialised; throws std::runtime_error
    ~DeviceArray();
    DeviceArray(DeviceArray&& other) noexcept;
    DeviceArray& operator=(DeviceArray&& other) noexcept;
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;

    double* data() const { return data_; }
    std::size_t size() const { return count_; }
    void upload(const std::vector<double>& host);     // resizes to host.size(); blocking
    std::vector<double> download() const;             // blocking; waits for the whole device first

   private:
    double* data_ = nullptr;
    std::size_t count_ = 0;
};

}  // namespace nbh
