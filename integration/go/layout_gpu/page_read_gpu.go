//go:build pqgpu

// GPU-backed read side of package layout: drop this file into github.com/xitongsys/parquet-go/layout.
//
// It defines the four read functions reader/columnbuffer.go calls - ReadPage (layout/page.go:721),
// ReadPageRawData (:395), (*Page).GetRLDLFromRawData (:433), (*Page).GetValueFromRawData (:562) - with their
// signatures, results and error values unchanged, so reader/columnbuffer.go and reader/columnreader.go compile and
// run as they are.  The CPU definitions of the same four functions move, verbatim, from layout/page.go into a file
// that carries `//go:build !pqgpu` (see INTEGRATION.md); nothing else in the repository changes.
// (*Page).Decode (:139) stays where it is: the pages handed out here already hold the looked-up values and say so
// in their header (Encoding PLAIN), which makes Decode return at its first test of the encoding.
//
// How it works: the reference reads and decodes one page per call.  Here the first call on a column chunk reads the
// whole chunk (ColumnMetaData.TotalCompressedSize bytes) from the thrift reader, has the GPU decode every page of it in
// one call of the C ABI and keeps the result; that call and the following ones each hand out one page of it, with the
// (numValues, numRows) the reference returns for that page.  The state is keyed by the thrift reader:
// ColumnBufferType opens a new one for every row group (reader/columnbuffer.go:95-99).
//
// Source only: written without a Go toolchain at hand; parquet-go_amd/layout_gpu.py makes the same calls in the same
// order and is what the tests drive.
package layout

import (
	"fmt"
	"io"
	"os"
	"runtime"
	"strconv"
	"sync"
	"unsafe"

	"github.com/apache/thrift/lib/go/thrift"
	"github.com/xitongsys/parquet-go/common"
	"github.com/xitongsys/parquet-go/parquet"
	"github.com/xitongsys/parquet-go/pqdecode"
	"github.com/xitongsys/parquet-go/schema"
)

// decoded is the result of one C-ABI call over the pages [first, last) of a chunk.
type decoded struct {
	first, last int
	lead        int // 1 when the chunk's dictionary page was put in front of the descriptors
	res         *pqdecode.Result
	slot0       []int // first slot of every page of the call, and the total
}

func newDecoded(first, last, lead int, res *pqdecode.Result) *decoded {
	d := &decoded{first: first, last: last, lead: lead, res: res, slot0: make([]int, last-first+1)}
	for k := first; k < last; k++ {
		d.slot0[k-first+1] = d.slot0[k-first] + int(res.Counts[lead+k-first].Slots)
	}
	return d
}

func (d *decoded) covers(k int) bool { return d != nil && d.first <= k && k < d.last }

type pageError struct {
	page int
	err  error
}

// chunkPager stands for the thrift reader positioned inside one column chunk.
type chunkPager struct {
	h        *pqdecode.Handle
	chunk    []byte // page-locked input buffer of the handle
	descs    []pqdecode.PageDesc
	col      pqdecode.Column
	cursor   int
	dictIdx  int // dictionary page handed out last, -1: none
	full     *decoded
	levels   *decoded
	errFull  *pageError
	errLevel *pageError

	path    []string
	repType parquet.FieldRepetitionType
	schema  *parquet.SchemaElement
	codec   parquet.CompressionCodec
}

var (
	pagers     sync.Map // uintptr(thrift reader) -> *chunkPager
	rawPages   sync.Map // uintptr(*Page handed out by ReadPageRawData) -> rawRef
	handlePool sync.Pool
	gpuDevice  = func() int { d, _ := strconv.Atoi(os.Getenv("PQD_DEVICE")); return d }()
)

type rawRef struct {
	pg    *chunkPager
	index int
}

func takeHandle() (*pqdecode.Handle, error) {
	if h, ok := handlePool.Get().(*pqdecode.Handle); ok && h != nil {
		return h, nil
	}
	return pqdecode.NewHandle(gpuDevice)
}

// pagerFor returns the hand-out state of the chunk the thrift reader is positioned in, reading the chunk on the
// first call.  io.EOF when not a byte is left, as ReadPageHeader reports it (reader/columnbuffer.go:108-129 gives
// that error a meaning).
func pagerFor(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*chunkPager, error) {
	key := uintptr(unsafe.Pointer(thriftReader))
	if p, ok := pagers.Load(key); ok {
		return p.(*chunkPager), nil
	}
	if colMetaData.GetCodec() != parquet.CompressionCodec_UNCOMPRESSED {
		return nil, fmt.Errorf("pqdecode: only UNCOMPRESSED column chunks are decoded on the GPU")
	}
	h, err := takeHandle()
	if err != nil {
		return nil, err
	}
	buf, err := h.InputBuffer(int(colMetaData.GetTotalCompressedSize()))
	if err != nil {
		handlePool.Put(h)
		return nil, err
	}
	n, err := io.ReadFull(thriftReader, buf)
	if n == 0 {
		handlePool.Put(h)
		if err == nil {
			err = io.EOF
		}
		return nil, err
	}
	buf = buf[:n] // a chunk cut short by the end of the file: the page that runs past it fails, not the chunk
	descs, err := pqdecode.ScanPages(buf, colMetaData.GetNumValues())
	if err != nil && len(descs) == 0 {
		handlePool.Put(h)
		return nil, err
	}
	path := append([]string{schemaHandler.GetRootInName()}, colMetaData.GetPathInSchema()...)
	name := common.PathToStr(path)
	maxDef, _ := schemaHandler.MaxDefinitionLevel(path)
	maxRep, _ := schemaHandler.MaxRepetitionLevel(path)
	el := schemaHandler.SchemaElements[schemaHandler.MapIndex[name]]
	pg := &chunkPager{h: h, chunk: buf, descs: descs, dictIdx: -1, path: path, repType: el.GetRepetitionType(), schema: el,
		codec: colMetaData.GetCodec(),
		col:   pqdecode.Column{Type: colMetaData.GetType(), TypeLength: el.GetTypeLength(), MaxDef: maxDef, MaxRep: maxRep}}
	pagers.Store(key, pg)
	// the map must not keep the reader alive: it is keyed by address, and the entry goes when the reader does
	runtime.SetFinalizer(thriftReader, func(t *thrift.TBufferedTransport) { pg.release(uintptr(unsafe.Pointer(t))) })
	return pg, nil
}

func (pg *chunkPager) release(key uintptr) {
	pagers.Delete(key)
	if pg.h != nil {
		handlePool.Put(pg.h)
		pg.h = nil
	}
}

// call decodes the pages [first, last), behind the dictionary page in effect when there is one.
func (pg *chunkPager) call(first, last int, levelsOnly bool) (*pqdecode.Result, int, error) {
	lead := 0
	descs := pg.descs[first:last]
	if pg.dictIdx >= 0 && !levelsOnly && pg.descs[first].PageType() != parquet.PageType_DICTIONARY_PAGE {
		lead = 1
		descs = append([]pqdecode.PageDesc{pg.descs[pg.dictIdx]}, descs...)
	}
	res, err := pg.h.Decode(pg.chunk, descs, pg.col, levelsOnly)
	return res, lead, err
}

// decodeFrom decodes the pages from k to the end of the chunk; pages in front of a failing one stay available.
func (pg *chunkPager) decodeFrom(k int, levelsOnly bool) error {
	last := len(pg.descs)
	res, lead, err := pg.call(k, last, levelsOnly)
	if err != nil {
		return err
	}
	var perr *pageError
	if res.Status != 0 {
		bad := k + res.BadPage - lead
		if lead == 1 && res.BadPage == 0 {
			bad = k // the dictionary page handed out earlier failed: the next page meets it
		}
		perr = &pageError{bad, pqdecode.Err(res.Status, bad)}
		last = bad
		if last > k {
			if res, lead, err = pg.call(k, last, levelsOnly); err != nil {
				return err
			}
			if res.Status != 0 {
				return pqdecode.Err(res.Status, k+res.BadPage-lead)
			}
		}
	}
	var d *decoded
	if last > k {
		d = newDecoded(k, last, lead, res)
	}
	if levelsOnly {
		pg.levels, pg.errLevel = d, perr
	} else {
		pg.full, pg.errFull = d, perr
		if perr == nil {
			pg.levels = d // a full decode without failures serves the level calls too
		}
	}
	return nil
}

// table builds the layout.Table of page k out of a call's result (layout/page.go:905-931).
func (pg *chunkPager) table(d *decoded, k int, withValues bool) (*Table, int64, int64) {
	a, b := d.slot0[k-d.first], d.slot0[k-d.first+1]
	t := new(Table)
	t.Path = pg.path
	t.RepetitionType = pg.repType
	t.MaxDefinitionLevel, t.MaxRepetitionLevel = pg.col.MaxDef, pg.col.MaxRep
	t.DefinitionLevels = d.res.Def[a:b:b]
	t.RepetitionLevels = d.res.Rep[a:b:b]
	if withValues && d.res.Values != nil {
		t.Values = d.res.Values[a:b:b]
	} else {
		t.Values = make([]interface{}, b-a)
	}
	return t, int64(b - a), d.res.Counts[d.lead+k-d.first].Rows
}

// header is the parquet.PageHeader of page k as far as the reader's callers look at it.  Encoding is PLAIN for every
// data page: its table holds values, never dictionary indices.
func (pg *chunkPager) header(k int) *parquet.PageHeader {
	d := &pg.descs[k]
	h := parquet.NewPageHeader()
	h.Type = d.PageType()
	h.CompressedPageSize, h.UncompressedPageSize = d.Length(), d.Length()
	switch d.PageType() {
	case parquet.PageType_DICTIONARY_PAGE:
		h.DictionaryPageHeader = parquet.NewDictionaryPageHeader()
		h.DictionaryPageHeader.NumValues = d.NumValues()
		h.DictionaryPageHeader.Encoding = parquet.Encoding_PLAIN
	default:
		h.DataPageHeader = parquet.NewDataPageHeader()
		h.DataPageHeader.NumValues = d.NumValues()
		h.DataPageHeader.Encoding = parquet.Encoding_PLAIN
		h.DataPageHeader.DefinitionLevelEncoding = parquet.Encoding_RLE
		h.DataPageHeader.RepetitionLevelEncoding = parquet.Encoding_RLE
	}
	return h
}

func (pg *chunkPager) newPage(k int) *Page {
	var page *Page
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		page = NewDictPage()
	} else {
		page = NewDataPage()
	}
	page.Header = pg.header(k)
	page.CompressType = pg.codec
	page.Path = pg.path
	page.Schema = pg.schema
	return page
}

func failed(e *pageError, k int) bool { return e != nil && k >= e.page }

// ReadPage hands out the next page of the chunk: (page, numValues, numRows, err) as layout/page.go:721-937 returns
// them.  A dictionary page comes back with an empty table (its entries are used on the device) and zeros.
func ReadPage(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*Page, int64, int64, error) {
	pg, err := pagerFor(thriftReader, schemaHandler, colMetaData)
	if err != nil {
		return nil, 0, 0, err
	}
	k := pg.cursor
	if k >= len(pg.descs) {
		return nil, 0, 0, io.EOF
	}
	switch pg.descs[k].PageType() {
	case parquet.PageType_DATA_PAGE, parquet.PageType_DATA_PAGE_V2, parquet.PageType_DICTIONARY_PAGE:
	case parquet.PageType_INDEX_PAGE:
		return nil, 0, 0, fmt.Errorf("Unsupported page type: INDEX_PAGE")
	default:
		return nil, 0, 0, fmt.Errorf("Error page type %v", pg.descs[k].PageType())
	}
	if !pg.full.covers(k) && !failed(pg.errFull, k) {
		if err = pg.decodeFrom(k, false); err != nil {
			return nil, 0, 0, err
		}
	}
	if failed(pg.errFull, k) {
		return nil, 0, 0, pg.errFull.err
	}
	pg.cursor++
	page := pg.newPage(k)
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		pg.dictIdx = k
		page.DataTable = &Table{Path: pg.path}
		return page, 0, 0, nil
	}
	var numValues, numRows int64
	page.DataTable, numValues, numRows = pg.table(pg.full, k, true)
	return page, numValues, numRows, nil
}

// ReadPageRawData hands out the next page undecoded (layout/page.go:395-430); GetRLDLFromRawData and
// GetValueFromRawData find their way back to the chunk through the page.
func ReadPageRawData(thriftReader *thrift.TBufferedTransport, schemaHandler *schema.SchemaHandler, colMetaData *parquet.ColumnMetaData) (*Page, error) {
	pg, err := pagerFor(thriftReader, schemaHandler, colMetaData)
	if err != nil {
		return nil, err
	}
	k := pg.cursor
	if k >= len(pg.descs) {
		return nil, io.EOF
	}
	switch pg.descs[k].PageType() {
	case parquet.PageType_DATA_PAGE, parquet.PageType_DATA_PAGE_V2, parquet.PageType_DICTIONARY_PAGE:
	default:
		return nil, fmt.Errorf("Unsupported page type")
	}
	pg.cursor++
	page := pg.newPage(k)
	key := uintptr(unsafe.Pointer(page))
	rawPages.Store(key, rawRef{pg, k})
	runtime.SetFinalizer(page, func(p *Page) { rawPages.Delete(uintptr(unsafe.Pointer(p))) })
	return page, nil
}

// GetRLDLFromRawData: levels, (numValues, numRows) of the page; values left undecoded (layout/page.go:433-559).
func (p *Page) GetRLDLFromRawData(schemaHandler *schema.SchemaHandler) (int64, int64, error) {
	ref, ok := rawPages.Load(uintptr(unsafe.Pointer(p)))
	if !ok {
		return 0, 0, fmt.Errorf("pqdecode: page was not handed out by ReadPageRawData")
	}
	pg, k := ref.(rawRef).pg, ref.(rawRef).index
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		pg.dictIdx = k // ReadPageForSkip keeps it as DictPage (reader/columnbuffer.go:170-174)
		p.DataTable = &Table{Path: pg.path}
		return 0, 0, nil
	}
	if !pg.levels.covers(k) && !failed(pg.errLevel, k) {
		if err := pg.decodeFrom(k, true); err != nil {
			return 0, 0, err
		}
	}
	if failed(pg.errLevel, k) {
		return 0, 0, pg.errLevel.err
	}
	var numValues, numRows int64
	p.DataTable, numValues, numRows = pg.table(pg.levels, k, false)
	return numValues, numRows, nil
}

// GetValueFromRawData: the values of a page whose levels are known (layout/page.go:562-622): the page alone, behind
// its dictionary page, in one call - ColumnBufferType.SkipRows asks for the last page it read only.
func (p *Page) GetValueFromRawData(schemaHandler *schema.SchemaHandler) error {
	ref, ok := rawPages.Load(uintptr(unsafe.Pointer(p)))
	if !ok {
		return fmt.Errorf("pqdecode: page was not handed out by ReadPageRawData")
	}
	pg, k := ref.(rawRef).pg, ref.(rawRef).index
	if pg.descs[k].PageType() == parquet.PageType_DICTIONARY_PAGE {
		return nil // its entries are decoded with the data pages that use them
	}
	d := pg.full
	if !d.covers(k) {
		res, lead, err := pg.call(k, k+1, false)
		if err != nil {
			return err
		}
		if res.Status != 0 {
			return pqdecode.Err(res.Status, k)
		}
		d = newDecoded(k, k+1, lead, res)
	}
	t, _, _ := pg.table(d, k, true)
	// keep the table GetRLDLFromRawData made (the caller has merged its levels already): fill in the values
	if p.DataTable == nil {
		p.DataTable = t
	} else {
		p.DataTable.Values = t.Values
	}
	p.RawData = []byte{}
	return nil
}
