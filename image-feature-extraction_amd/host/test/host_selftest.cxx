// host_selftest -- exercises the host mirror end to end on a GPU box: file IO round trip,
// the filter classes (same call sequence as the reference's tools), the functors on the
// reference's seven known answers (test/Symmetric3x3EigenvalueSolverTest.cxx:48-90, as data
// in tests/golden/eigen_kat.json), and error translation.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ife/Filters/Hessian3DImageFilter.h"
#include "ife/Filters/ImageToEmphysemaFeaturesFilter.h"
#include "ife/Filters/NormalizedGaussianConvolutionImageFilter.h"
#include "ife/Host/ImageIO.h"
#include "ife/Host/LiteFilters.h"
#include "ife/IO/IO.h"
#include "ife/Statistics/DenseHistogram.h"
#include "ife/Statistics/DetermineEdgesForEqualizedHistogram.h"
#include "ife/Numerics/EigenvalueFeaturesFunctor.h"
#include "ife/Util/Path.h"

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);          \
      ++failures;                                                          \
    }                                                                      \
  } while (0)

typedef itk::Image<float, 3> ImageType;
typedef itk::Image<unsigned char, 3> MaskType;
typedef itk::VectorImage<float, 3> VectorImageType;

int main(int argc, char **argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp";
  CHECK(Path::join("a/b//", "/c") == "a/b/c");
  CHECK(Path::join("out", "hessian_") == "out/hessian_");
  CHECK(Path::join("///", "x") == "/x");

  // a small volume with structure
  itk::Size3 sz; sz[0] = 20; sz[1] = 12; sz[2] = 9;
  ImageType::Pointer img = ImageType::New();
  img->SetRegions(sz);
  itk::Spacing3 sp; sp[0] = 0.5; sp[1] = 1.0; sp[2] = 2.0;
  img->SetSpacing(sp);
  img->Allocate();
  MaskType::Pointer mask = MaskType::New();
  mask->CopyInformation(img);
  mask->Allocate();
  for (unsigned z = 0; z < sz[2]; ++z)
    for (unsigned y = 0; y < sz[1]; ++y)
      for (unsigned x = 0; x < sz[0]; ++x) {
        const size_t i = x + sz[0] * (y + sz[1] * z);
        img->GetBufferPointer()[i] = (float)((x * x * 3 + y * y * 5 + z * 7 + x * y) % 97) - 40.0f;
        mask->GetBufferPointer()[i] = (unsigned char)((x > 2 && y > 1) ? ((x + y + z) % 5 == 0 ? 2 : 1) : 0);
      }

  try {
    // IO round trips (.nii.gz, .nii, .mhd), including type conversion on read
    const char *ext[3] = {".nii.gz", ".nii", ".mhd"};
    for (int k = 0; k < 3; ++k) {
      const std::string f = tmp + "/ife_selftest_img" + ext[k];
      itk::ImageFileWriter<ImageType>::Pointer w = itk::ImageFileWriter<ImageType>::New();
      w->SetInput(img);
      w->SetFileName(f);
      w->Update();
      itk::ImageFileReader<ImageType>::Pointer r = itk::ImageFileReader<ImageType>::New();
      r->SetFileName(f);
      ImageType *back = r->GetOutput();
      CHECK(back->GetLargestPossibleRegion().GetSize()[0] == 20);
      CHECK(back->GetSpacing()[2] == 2.0);
      bool same = true;
      for (size_t i = 0; i < 20 * 12 * 9; ++i) same = same && back->GetBufferPointer()[i] == img->GetBufferPointer()[i];
      CHECK(same);
      itk::ImageFileReader<itk::Image<double, 3> >::Pointer rd = itk::ImageFileReader<itk::Image<double, 3> >::New();
      rd->SetFileName(f);
      CHECK(rd->GetOutput()->GetBufferPointer()[5] == (double)img->GetBufferPointer()[5]);
    }

    // feature filter: clamp the labels like ExtractFeatures.cxx:99-104, two scales
    itk::ClampImageFilter<MaskType, MaskType>::Pointer clamp = itk::ClampImageFilter<MaskType, MaskType>::New();
    clamp->SetBounds(0, 1);
    clamp->SetInput(mask);
    typedef itk::ImageToEmphysemaFeaturesFilter<ImageType, MaskType, VectorImageType> FeatureFilterType;
    FeatureFilterType::Pointer ff = FeatureFilterType::New();
    CHECK(FeatureFilterType::numFeatures == 8);
    CHECK(ff->GetSigma() == 1.0f);
    ff->SetInputImage(img);
    ff->SetInputMask(clamp->GetOutput());
    ff->SetSigma(1.5f);
    ff->UpdateLargestPossibleRegion();
    VectorImageType *fo = ff->GetOutput();
    CHECK(fo->GetNumberOfComponentsPerPixel() == 8);
    const float *f = fo->GetBufferPointer();
    bool masked_zero = true, some_nonzero = false, ordered = true;
    for (size_t i = 0; i < 20 * 12 * 9; ++i) {
      const bool in = clamp->GetOutput()->GetBufferPointer()[i] != 0;
      for (int c = 0; c < 8; ++c) {
        if (!in) masked_zero = masked_zero && f[i * 8 + c] == 0.0f;
        else some_nonzero = some_nonzero || f[i * 8 + c] != 0.0f;
      }
      if (in) ordered = ordered && std::fabs(f[i * 8 + 2]) >= std::fabs(f[i * 8 + 3]) &&
                        std::fabs(f[i * 8 + 3]) >= std::fabs(f[i * 8 + 4]);
    }
    CHECK(masked_zero); CHECK(some_nonzero); CHECK(ordered);
    const float keep = f[8 * 777 + 2];
    ff->Update();  // unchanged inputs: no re-execution, same buffer
    CHECK(ff->GetOutput()->GetBufferPointer()[8 * 777 + 2] == keep);

    // component selection + Hessian filter + functor consistency
    itk::Hessian3DImageFilter<ImageType, VectorImageType>::Pointer hf =
        itk::Hessian3DImageFilter<ImageType, VectorImageType>::New();
    hf->SetInput(img);
    hf->Update();
    CHECK(hf->GetOutput()->GetNumberOfComponentsPerPixel() == 6);
    EigenvalueFeaturesFunctor<float> functor;
    const size_t probe = 5 + 20 * (6 + 12 * 4);
    itk::VariableLengthVector<float> A(hf->GetOutput()->GetBufferPointer() + probe * 6, 6);
    itk::VariableLengthVector<float> ef = functor(A);
    CHECK(ef.Size() == 6);
    CHECK(ef[3] == ef[0] + ef[1] + ef[2]);
    CHECK(functor == EigenvalueFeaturesFunctor<float>());

    // the reference's known answers through the host functor
    const float kat[7][6] = {{1, 0, 0, 1, 0, 1}, {1, 0, 0, 2, 0, 3}, {-1, 0, 0, -2, 0, -3}, {1, 0, 0, -2, 0, 3},
                             {1, 1, 1, 1, 1, 1}, {0.27f, 0.92f, 0.58f, 0.24f, 0.75f, 0.04f},
                             {599, 860, -835, -941, 817, -207}};
    const double exp[7][3] = {{1, 1, 1}, {3, 2, 1}, {-3, -2, -1}, {3, -2, 1}, {3, 0, 0},
                              {1.70680634, -0.7205504, -0.43625594},
                              {-2005.21004566, 1183.41690727, 272.79313839}};
    Symmetric3x3EigenvalueSolver<float> solver;
    for (int k = 0; k < 7; ++k) {
      itk::VariableLengthVector<float> ev = solver(itk::VariableLengthVector<float>(kat[k], 6));
      double scale = 1.0;
      for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(exp[k][c]));
      for (int c = 0; c < 3; ++c) CHECK(std::fabs(ev[c] - exp[k][c]) <= 2e-6 * scale);
    }

    // normalized convolution: certainty one everywhere keeps a constant image constant
    ImageType::Pointer ones = ImageType::New();
    ones->CopyInformation(img);
    ones->Allocate();
    ImageType::Pointer cst = ImageType::New();
    cst->CopyInformation(img);
    cst->Allocate();
    for (size_t i = 0; i < 20 * 12 * 9; ++i) { ones->GetBufferPointer()[i] = 1.0f; cst->GetBufferPointer()[i] = 7.5f; }
    itk::NormalizedGaussianConvolutionImageFilter<ImageType>::Pointer nc =
        itk::NormalizedGaussianConvolutionImageFilter<ImageType>::New();
    nc->SetInputImage(cst);
    nc->SetInputCertainty(ones);
    nc->SetSigma(2.0);
    nc->Update();
    bool all75 = true;
    for (size_t i = 0; i < 20 * 12 * 9; ++i) all75 = all75 && nc->GetOutput()->GetBufferPointer()[i] == 7.5f;
    CHECK(all75);

    // error translation: an axis shorter than 4 is rejected like ITK does
    itk::Size3 tiny; tiny[0] = 8; tiny[1] = 8; tiny[2] = 3;
    ImageType::Pointer ti = ImageType::New(); ti->SetRegions(tiny); ti->Allocate();
    MaskType::Pointer tm = MaskType::New(); tm->SetRegions(tiny); tm->Allocate();
    FeatureFilterType::Pointer bad = FeatureFilterType::New();
    bad->SetInputImage(ti); bad->SetInputMask(tm);
    bool threw = false;
    try { bad->Update(); } catch (itk::ExceptionObject &e) { threw = std::string(e.what()).find("at least 4") != std::string::npos; }
    CHECK(threw);

    // rows f1 / f2 through the reference's own interfaces, on its own known answers
    // (test/DetermineEdgesForEqualizedHistogramTest.cxx:30-70, test/DenseHistogramTest.cxx:10-55)
    {
      std::vector<double> values{1, 2, 3, 4, 5, 6, 7, 8, 9}, edges(2);
      determineEdgesForEqualizedHistogram(values.begin(), values.end(), edges.begin(), 3);
      CHECK(edges[0] == 4 && edges[1] == 7);
      std::vector<double> same(8, 1), e1{0, 123};
      determineEdgesForEqualizedHistogram(same.begin(), same.end(), e1.begin(), 2);
      CHECK(e1[0] == 1 && e1[1] == 123);
      std::vector<float> uneven{1, 1, 1, 1, 1, 2, 2, 3, 3, 3}, e2;
      determineEdgesForEqualizedHistogram(uneven.begin(), uneven.end(), std::back_inserter(e2), 3);
      CHECK(e2.size() == 2 && e2[0] == 2 && e2[1] == 3);
      bool range = false;
      std::vector<double> e9(9);
      try { determineEdgesForEqualizedHistogram(values.begin(), values.end(), e9.begin(), 10); }
      catch (const std::out_of_range &) { range = true; }
      CHECK(range);

      const float vals[18] = {-1, 0, 0.5f, 1, 1.5f, 2.1f, 2.6f, 2.9f, 3.2f, 3.5f, 4.2f, 4.6f, 5, 6, 7, 8, 9, 10};
      DenseHistogram<float> hist({1, 2.5f, 3.0f, 4.7f, 6.2f, 8.3f});
      for (float v : vals) hist.insert(v);
      const unsigned int expected[7] = {4, 2, 2, 4, 2, 2, 2};
      const std::vector<unsigned int> counts = hist.getCounts();
      const std::vector<float> freq = hist.getFrequencies();
      CHECK(counts.size() == 7 && hist.getNumberOfBins() == 7);
      for (size_t i = 0; i < 7 && i < counts.size(); ++i) {
        CHECK(counts[i] == expected[i]);
        CHECK(std::fabs(freq[i] - expected[i] / 18.0f) < 1e-7f);
      }
      std::ostringstream os;
      os << hist;
      CHECK(os.str() == "4,2,2,4,2,2,2");
      hist.resetCounts();
      hist.insert(100.0f);
      CHECK(hist.getCounts()[6] == 1 && hist.getCounts()[0] == 0);

      std::ostringstream row;
      const float fr[6] = {1.5f, -0.0f, 1e-7f, 123456789.0f, 3.1415927f, 1e10f};
      writeSequenceAsText(row, fr, fr + 6);
      CHECK(row.str() == "1.5,-0,1e-07,1.23457e+08,3.14159,1e+10");  // tests/golden/stats_ref.json
      const std::string list = tmp + "/ife_selftest_pairs.csv";
      { std::ofstream f(list.c_str()); f << "a.nii.gz, m a.nii.gz \n\n b.nii,b_mask.nii\r\n"; }
      const std::vector<StringPair> pairs = readPairList(list);
      CHECK(pairs.size() == 2 && pairs[0].first == "a.nii.gz" && pairs[0].second == "m a.nii.gz");
      CHECK(pairs.size() == 2 && pairs[1].first == "b.nii" && pairs[1].second == "b_mask.nii");
      { std::ofstream f(list.c_str()); f << "no separator here\n"; }
      bool inv = false;
      try { readPairList(list); } catch (const std::invalid_argument &) { inv = true; }
      CHECK(inv);
    }
  } catch (itk::ExceptionObject &e) {
    std::cout << "unexpected " << e << std::endl;
    return 2;
  }
  std::printf(failures ? "host_selftest: %d failure(s)\n" : "host_selftest: ok\n", failures);
  return failures ? 1 : 0;
}
