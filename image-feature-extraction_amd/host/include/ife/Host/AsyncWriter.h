// AsyncWriter.h -- writes images on worker threads so that the (zlib-bound) compression of
// one scale's eight .nii.gz files runs beside the device work of the next scale
// (SURVEY.md section 8 row f3).  The reference writes serially from the ITK pipeline
// (tools/ExtractFeatures.cxx:135-143); file names and contents are unchanged.
//
//   ife::host::AsyncWriter<ImageType> writers;       // IFE_WRITER_THREADS, default min(8, cores)
//   writers.Submit(image, path);                     // takes shared ownership of the image
//   writers.Wait();                                  // rethrows the first failure
//
// At most `max_pending` images are held at a time; Submit blocks beyond that.
#ifndef IFE_HOST_ASYNC_WRITER_H
#define IFE_HOST_ASYNC_WRITER_H

#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ife/Host/ImageIO.h"

namespace ife {
namespace host {

template <typename TImage>
class AsyncWriter {
 public:
  explicit AsyncWriter(size_t max_pending = 16) : max_pending_(max_pending) {
    unsigned n = std::thread::hardware_concurrency();
    n = n == 0 ? 1 : (n > 8 ? 8 : n);
    if (const char *e = std::getenv("IFE_WRITER_THREADS")) n = (unsigned)std::atoi(e);
    for (unsigned i = 0; i < (n == 0 ? 1u : n); ++i) workers_.emplace_back([this] { Run(); });
  }
  ~AsyncWriter() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread &t : workers_) t.join();
  }
  void Submit(const typename TImage::Pointer &image, const std::string &path) {
    std::unique_lock<std::mutex> l(m_);
    space_.wait(l, [this] { return queue_.size() + active_ < max_pending_; });
    queue_.push_back(Job{image, path});
    cv_.notify_one();
  }
  // all submitted files are on disk afterwards; throws the first error a worker met
  void Wait() {
    std::unique_lock<std::mutex> l(m_);
    space_.wait(l, [this] { return queue_.empty() && active_ == 0; });
    if (!error_.empty()) {
      const std::string e = error_;
      error_.clear();
      throw itk::ExceptionObject(e, "AsyncWriter");
    }
  }

 private:
  struct Job {
    typename TImage::Pointer image;
    std::string path;
  };
  void Run() {
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;
        job = queue_.front();
        queue_.pop_front();
        ++active_;
      }
      std::string err;
      try {
        typename itk::ImageFileWriter<TImage>::Pointer w = itk::ImageFileWriter<TImage>::New();
        w->SetInput(job.image.GetPointer());
        w->SetFileName(job.path);
        w->Update();
      } catch (std::exception &e) {
        err = job.path + ": " + e.what();
      }
      job.image = typename TImage::Pointer();
      {
        std::lock_guard<std::mutex> l(m_);
        --active_;
        if (!err.empty() && error_.empty()) error_ = err;
      }
      space_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_, space_;
  std::deque<Job> queue_;
  std::vector<std::thread> workers_;
  size_t max_pending_, active_ = 0;
  bool stop_ = false;
  std::string error_;
};

}  // namespace host
}  // namespace ife

#endif
